#!/bin/bash
# One GPU-box call of a development round: the -m gpu suite, driver-form bench lines, an A/B of variant libraries against the
# product, the parity soak.  Usage: tools/gpu_round.sh <tag> "<variants for ab_headline>" [soak args]
tag=${1:-round}; variants=${2:-}; soak=${3:-}
out=gpurun_out/$tag; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
tools/bench_driver_form.sh $tag 3
if [ -n "$variants" ]; then tools/ab_headline.sh $variants | tee $out/ab.txt; fi
python3 tools/soak_parity.py $soak > $out/soak.txt 2>&1; tail -8 $out/soak.txt
