#!/bin/bash
# round 5: config-4 tests + timings with the memo of the full dyn step on / off, and per-env worlds (no sharing possible)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r5c4}
mkdir -p $OUT
cd $ROOT
timeout 1200 python3 -m pytest tests/test_config4_gpu.py -x -q > $OUT/pytest_c4.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_c4.log
for M in 1 0; do MEMO=$M timeout 300 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-420 | tee -a $OUT/c4_time.txt; done
SSG_DYN_STOP=-1 python3 tools/c4_stamps_last.py 2>&1 | tail -7 | cut -c1-400 | tee $OUT/stamps.txt
if [ -n "$FRESH" ]; then MAP_MODE=fresh_device RING=8 K=100 timeout 600 python3 tools/time_config4.py 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/c4_time.txt; fi
