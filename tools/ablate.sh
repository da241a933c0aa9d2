#!/bin/bash
# timing-only ablation sweep (needs a -DSSG_ABLATION build); prints avg launch us per variant
for A in ${ABLS:-0 0x3F 0x7F 0xBF 0x13F 0x1FF 0x40 0x80 0x100}; do
  SSG_ABLATE=$A python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate=$A', round(d['roofline']['avg_launch_us'],2), 'us')"
done
