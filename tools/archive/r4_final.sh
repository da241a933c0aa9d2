#!/bin/bash
# round-4 closing GPU call: bench lines of the final tree, config-4 kernel times, the long parity soak
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4final
mkdir -p $OUT
cd $ROOT
timeout 900 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench default rc=$?"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_style_steps20.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
( cd /tmp && export TMPDIR=/tmp && K=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4trace -- python3 $ROOT/tools/time_config4.py > $OUT/c4_trace.log 2>&1 )
cp $OUT/c4trace/*/*kernel_stats.csv $OUT/c4_kernel_stats.csv 2>/dev/null
find $OUT/c4trace -name "*kernel_trace.csv" -size +3M -delete
timeout 300 python3 tools/time_config4.py > $OUT/c4_time.txt 2>&1
timeout 2400 python3 tools/soak_parity.py --long > $OUT/soak_parity.txt 2>&1; echo "soak rc=$?"
tail -n 12 $OUT/soak_parity.txt
python3 - <<PY
import json
for f in ("bench_line.json", "bench_line_driver_style_steps20.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("repeats_ms"), d["roofline"].get("frac"), d["roofline"].get("traffic"), d.get("single_step_launch_us"))
    for k, v in (d.get("other_configs") or {}).items():
        print("   ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("us_per_step", "env_steps_per_s", "frac")} if isinstance(v, dict) else v)
PY
