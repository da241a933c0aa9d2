#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python3 -m pytest tests/test_fresh_device_gpu.py tests/test_parity_gpu.py -x -q -k "fresh or gathered or bank_in_global or lds_fit or ragged" 2>&1 | tail -3
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
dev = torch.device("cuda", 0)
for ring, K, W in ((128, 635, 127), (48, 470, 94)):
    r = bench.side_config(dev, 65536, 8, 1, K, W, map_mode="fresh_device", ring=ring)
    print("fresh_device ring %d: %.3f us per step" % (ring, r["us_per_step"]))
PY
