cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 tools/gather_locality_probe.py
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
dev = torch.device("cuda", 0)
r = bench.side_config(dev, 65536, 8, 1, 635, 127, map_mode="fresh_device", ring=128)
print("c3 fresh_device ring 128: %.3f us per step" % r["us_per_step"])
r = bench.side_config(dev, 65536, 10, 4, 62, 31, map_mode="fresh_device", ring=32)
print("c4 fresh_device ring 32: %.3f us per step" % r["us_per_step"])
PY
timeout 1200 python3 -m pytest tests/test_fresh_device_gpu.py tests/test_parity_gpu.py -x -q -k "fresh or gathered or bank_in_global or lds_fit or ragged or geometry or block" 2>&1 | tail -3
