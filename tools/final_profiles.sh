#!/bin/bash
# The judged measurements of a round, taken ONCE on its final tree (GPU box):  tools/final_profiles.sh <round tag, e.g. r6>
#  1. headline kernel: rocprofv3 kernel trace + PMC passes + FETCH / WRITE calibration (tools/profile_bench.sh <tag>h)
#  2. config 4 with the memo (trace + PMC) and without it (trace), per-launch times of the full dyn step
#  3. the un-profiled bench lines (default form and the driver's --steps 20 --warmup 5 form, three of each in fresh processes)
#  4. the parity soak (SOAK=--long for the extended one)
# Copy the summaries into profiles/<tag>/ afterwards with tools/install_profiles.sh <tag>.
R=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/profile_bench.sh ${R}h > gpurun_out/prof_${R}h.log 2>&1
K=300 bash tools/profile_c4.sh $R > gpurun_out/prof_c4_$R.log 2>&1
( cd /tmp && export TMPDIR=/tmp && MEMO=0 K=200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c4_$R/trace_memo_off -- python3 $ROOT/tools/time_config4.py > $ROOT/gpurun_out/prof_c4_$R/trace_memo_off.log 2>&1 )
K=300 bash tools/c4_kernel_times.sh product > gpurun_out/prof_c4_$R/kernel_times.txt 2>&1
find gpurun_out/prof_c4_$R gpurun_out/prof_${R}h -name "*.csv" -size +3M -delete
OUT=gpurun_out/${R}final; mkdir -p $OUT
for i in 1 2 3; do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_form_$i.json 2> $OUT/bench_driver_$i.err; echo "bench driver form $i rc=$?"
done
timeout 1500 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench default rc=$?"
timeout 2400 python3 tools/soak_parity.py ${SOAK:-} > $OUT/soak_parity.txt 2>&1; echo "soak rc=$?"
tail -n 12 $OUT/soak_parity.txt
tail -40 gpurun_out/prof_${R}h.log
