#!/bin/bash
# round 5: the judged profiles — headline kernel (kernel trace + PMC passes + calibration), config 4 with the memo (trace + PMC) and
# without it (trace), per-launch times of the full dyn step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/profile_bench.sh r5h > gpurun_out/prof_r5h.log 2>&1
K=300 bash tools/profile_c4.sh r5 > gpurun_out/prof_c4_r5.log 2>&1
( cd /tmp && export TMPDIR=/tmp && MEMO=0 K=200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c4_r5/trace_memo_off -- python3 $ROOT/tools/time_config4.py > $ROOT/gpurun_out/prof_c4_r5/trace_memo_off.log 2>&1 )
K=300 bash tools/c4_kernel_times.sh product > gpurun_out/prof_c4_r5/kernel_times.txt 2>&1
find gpurun_out/prof_c4_r5 gpurun_out/prof_r5h -name "*.csv" -size +3M -delete
tail -30 gpurun_out/prof_c4_r5.log; tail -5 gpurun_out/prof_c4_r5/kernel_times.txt
