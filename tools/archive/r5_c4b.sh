#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r5c4b}
mkdir -p $OUT
cd $ROOT
MEMO=1 timeout 300 python3 tools/time_config4.py 2>&1 | tail -2 | tee -a $OUT/c4_time.txt
