#!/bin/bash
# Development aid (GPU box): the DRIVER-form line (--steps 20 --warmup 5) of the product library beside variant libraries, interleaved.
for i in 1 2 3; do
for v in product "$@"; do
  if [ $v = product ]; then unset SSG_LIB_PATH; else export SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_$v.so; fi
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-single-step --no-other-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e9,3), 'G', ['%.4f'%x for x in d['repeats_ms']], ['%.4f'%x for x in d['repeats_event_ms']], '%.2f GHz'%d['repeats_shader_clock_ghz'][2])"
done; done
