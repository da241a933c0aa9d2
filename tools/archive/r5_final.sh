#!/bin/bash
# round-5 closing GPU call: bench lines of the final tree (with the sha-tied PMC profile installed), the parity soak
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5final
mkdir -p $OUT
cd $ROOT
timeout 1200 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench default rc=$?"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_style_steps20.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
timeout 300 python3 tools/time_config4.py > $OUT/time_config4.txt 2>&1
timeout 1800 python3 tools/soak_parity.py ${SOAK:-} > $OUT/soak_parity.txt 2>&1; echo "soak rc=$?"
tail -n 12 $OUT/soak_parity.txt
python3 - <<PY
import json
for f in ("bench_line.json", "bench_line_driver_style_steps20.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], d.get("repeats_ms"), r.get("frac"), r.get("frac_wall"), r.get("frac_fused_compulsory"), r.get("bound"), r.get("traffic"), d.get("single_step_launch_us"))
    for k, v in (d.get("other_configs") or {}).items():
        print("   ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("us_per_step", "env_steps_per_s", "frac", "kernel_split_us")} if isinstance(v, dict) else v)
PY
