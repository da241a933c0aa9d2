#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + a few PMC passes over tools/time_config4.py (config 4).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_c4_${1:-r1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export K=${K:-200}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/time_config4.py > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $ROOT/tools/time_config4.py > $OUT/pmc$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats ==")
    for r in list(csv.DictReader(open(f)))[:8]:
        print("%-70s calls %6s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for d in sorted(glob.glob(out + "/pmc*")):
    if not d.endswith(tuple("0123456789")): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dyn_step" in k or "step_kernel" in k:
                acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: sum(v) / len(v) for c, v in cs.items()})
PY
find $OUT -name "*.csv" -size +3M -delete
