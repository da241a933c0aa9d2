#!/bin/bash
# round-4 GPU call 1: suite on the housekeeping commit, fresh config-4 anatomy, driver-style line of this box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4a
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
N=65536 SSG_DYN_STOP=-1 timeout 300 python3 tools/c4_stamps.py > $OUT/c4_stamps.txt 2>&1
N=65536 SSG_DYN_STOP=-1 SSG_LIB_PATH=$ROOT/ship_sim_gym_amd/libshipsim_dynprof.so timeout 300 python3 tools/c4_stamps.py > $OUT/c4_stamps_prof.txt 2>&1
timeout 300 python3 tools/time_config4.py > $OUT/c4_time.txt 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
tail -40 $OUT/c4_stamps_prof.txt
tail -3 $OUT/c4_time.txt
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("driver", d["value"], d["ms_per_step"], d["repeats_ms"], d.get("single_step_launch_us"))
for k,v in (d.get("other_configs") or {}).items():
    print("   ",k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("us_per_step","env_steps_per_s","frac")} if isinstance(v,dict) else v)
PY
