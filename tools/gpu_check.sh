#!/bin/bash
# Runs on the GPU box (via gpurun): the -m gpu suite, then the two bench lines (default and driver-style).
# Usage: tools/gpu_check.sh <tag> [pytest args]    -> gpurun_out/<tag>/
TAG=${1:-chk}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
python3 - <<PY
import json
for f in ("bench_default","bench_driver"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "value %.4g"%d["value"], "ms/step %.5f"%d["ms_per_step"], "frac %.3f"%r["frac"], "us/step-in-launch %.3f"%r["us_per_step_in_launch"], "single", d.get("single_step_launch_us"))
        for k,v in (d.get("other_configs") or {}).items():
            print("   ",k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("us_per_step","env_steps_per_s","frac")} if isinstance(v,dict) else v)
        print("    cpu", d.get("cpu_baseline"))
    except Exception as e:
        print(f, "unreadable", e); print(open("$OUT/%s.err"%f).read()[-2000:])
PY
