#!/bin/bash
# Development aid (GPU box): which shader clock do the timed repeats run at, against steps per repeat, number of repeats and the length
# of the conditioning?  (Round 6: 2.12-2.15 GHz for 20-step repeats whatever the conditioning — 300 ms .. 3 s, launch-and-wait or back
# to back, 20- or 100-step launches; 2.15 -> 2.41 over the first two 2 000-step repeats; 2.15 -> 1.85 over five 100-step repeats; 1.74 ->
# 2.19 over five 500-step repeats: the governor answers a change of duty with an excursion of tens of ms.)
for cfg in "20 5 300" "100 5 300" "500 5 300" "2000 5 300" "20 60 300" "20 5 1500"; do
  set -- $cfg
  python3 bench.py --gpus 1 --steps $1 --warmup 5 --repeats $2 --precondition-ms $3 --no-cpu-baseline --no-single-step --no-other-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['repeats_shader_clock_ghz']; print('steps $1 repeats $2 conditioning $3 ms:', round(d['value']/1e9,3), 'G', 'us/step', ['%.2f'%(x*1e3/$1) for x in d['repeats_ms']][:12], 'clk', ['%.2f'%x for x in c][:12], '... last', '%.2f'%c[-1])"
done
