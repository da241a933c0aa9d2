#!/bin/bash
# Development aid (GPU box): config 4 on the per-lane-planes kernel with 32 (product) / 16 / 8 envs per wave
# (tools/build_variant.sh g16 -DSSG_DYN_NONUNI_GRP=16, g8 likewise).
for v in product "$@"; do
  if [ $v = product ]; then unset SSG_LIB_PATH; else export SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_$v.so; fi
  for cfg in "MAP_MODE=fresh_device RING=32 K=124" "MAPS=96 K=300" "MAPS=96 MEMO=0 K=300"; do
    echo "== $v $cfg: $(env $cfg python3 tools/time_config4.py 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f us/step'%j['us_per_step'])")"
  done
done
