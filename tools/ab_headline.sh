#!/bin/bash
# Development aid (GPU box): the headline line of the product library beside variant libraries (tools/build_variant.sh <tag>),
# interleaved:   tools/ab_headline.sh base nmA ...
for i in 1 2 3; do
for v in product "$@"; do
  if [ $v = product ]; then unset SSG_LIB_PATH; else export SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_$v.so; fi
  python3 bench.py --no-cpu-baseline --no-single-step --no-other-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e9,3), 'G', round(d['ms_per_step']*1e3,3), 'us/step')"
done; done
