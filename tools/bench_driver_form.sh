#!/bin/bash
# Driver-form bench lines (`--gpus 1 --steps 20 --warmup 5`), N runs back to back in fresh processes, plus one without the
# device conditioning for comparison.  Usage (GPU box): tools/bench_driver_form.sh <tag> [N]
set -u
tag=${1:-driver_form}; n=${2:-3}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $out/line_$i.json 2> $out/err_$i.txt
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --precondition-ms 0 > $out/line_nocond.json 2> $out/err_nocond.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/line_*.json")):
    try:
        j=json.load(open(f))
        print(f, "value %.3f G"%(j["value"]/1e9), "repeats_ms", ["%.4f"%x for x in j["repeats_ms"]], "clk", ["%.2f"%x for x in j["repeats_shader_clock_ghz"]],
              "ev", ["%.4f"%x for x in j["repeats_event_ms"]], "pre %.0f ms"%j["preconditioning_ms"], "single %.2f"%(j.get("single_step_launch_us") or 0))
    except Exception as ex:
        print(f, "ERR", ex)
PY
