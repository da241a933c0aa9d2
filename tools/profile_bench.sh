#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes over a short bench.py run.
# Usage: tools/profile_bench.sh <tag> [bench args...]    -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2000 --warmup 200 --no-cpu-baseline --no-single-step --no-other-configs $*"  # every step_kernel dispatch is a 100-step launch of the headline kernel
# 1. kernel trace + stats (no counters): the same command as the bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
# 2..n: PMC passes, each in its own run, kernel-trace only alongside
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
# calibration of FETCH_SIZE / WRITE_SIZE on a known 8-byte-per-lane byte count
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $ROOT/tools/calib_pmc.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- python3 $ROOT/tools/calib_pmc.py > $OUT/cal_write.log 2>&1
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +3M -delete
tail -60 $OUT/summary.txt
