#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/ab_headline.sh "$@"
for v in product "$@"; do
  if [ $v = product ]; then unset SSG_LIB_PATH; else export SSG_LIB_PATH=$PWD/ship_sim_gym_amd/libshipsim_$v.so; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v driver-form', round(d['value']/1e9,3), 'G', round(d['ms_per_step']*1e3,3), 'us/step, single', round(d['single_step_launch_us'],2))"
done
unset SSG_LIB_PATH
SSG_DYN_STOP=-1 python3 tools/c4_stamps_last.py 2>&1 | tail -7 | cut -c1-420
for i in 1 2; do MEMO=1 timeout 300 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-120; done
