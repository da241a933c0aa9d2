#!/bin/bash
# Development aid (GPU box): one-launch-per-step times (ssg_step) of the product library and of variant libraries, interleaved:
# 65 536 envs x 8 beams and 4 096 x 10, then config 4 (memo on / off).   tools/single_step_ab.sh <variant tags...>
for i in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then unset SSG_LIB_PATH; else export SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_$v.so; fi
  echo "$v 65536x8: $(python3 tools/single_step_probe.py 2>/dev/null | tail -1)"
  echo "$v 4096x10: $(SSG_N=4096 SSG_NB=10 python3 tools/single_step_probe.py 2>/dev/null | tail -1)"
done; done
for v in product "$@"; do
  if [ $v = product ]; then unset SSG_LIB_PATH; else export SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_$v.so; fi
  for cfg in "K=300" "MEMO=0 K=300"; do
    echo "== $v c4 $cfg: $(env $cfg python3 tools/time_config4.py 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f us/step'%j['us_per_step'])")"
  done
done
