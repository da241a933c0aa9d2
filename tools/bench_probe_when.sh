#!/bin/bash
# Development aid (GPU box): does the (light) shader-clock probe after a timed repeat disturb the next repeat?
for i in 1 2 3; do
for w in after off; do
  SSG_BENCH_CLOCKPROBE=$w python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-single-step 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$w', 'value %.3f G'%(j['value']/1e9), ['%.4f'%x for x in j['repeats_ms']], ['%.4f'%x for x in j['repeats_event_ms']], [('%.2f'%x if x else None) for x in j['repeats_shader_clock_ghz']], 'pre %.0f ms / %d'%(j['preconditioning_ms'], 0))"
done; done
