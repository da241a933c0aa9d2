#!/bin/bash
# Development aid (GPU box): memory-path counters of the gathered step kernel on shared / dense / ring records (tools/gather_locality_probe.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/gather_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1 || rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "TCP_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" $OUT/avail.txt | sort -u > $OUT/names.txt
wc -l $OUT/names.txt
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "TCC_EA_RDREQ_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_TOTAL_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/gp$i
  K=200 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/gp$i -- python3 $ROOT/tools/gather_locality_probe.py > /tmp/gp$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 - <<'PY' | tee $OUT/summary.txt
import csv,glob,collections
for i in range(1,9):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    order=[]
    for f in glob.glob("/tmp/gp%d/**/*counter_collection.csv" % i, recursive=True):
        rows=list(csv.DictReader(open(f)))
        rows.sort(key=lambda r:int(r["Dispatch_Id"]))
        # the probe runs shared, ring, dense1, dense4 in this order: split the step-kernel dispatches by the reset kernels between them
        phase=0; last_reset=False
        for r in rows:
            k=r["Kernel_Name"]
            if "reset_kernel" in k and "dyn" not in k:
                if not last_reset: phase+=1
                last_reset=True; continue
            if "step_kernel" in k:
                last_reset=False
                acc[phase][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for ph in sorted(acc):
        print("pass", i, "phase", ph, {a:(round(sum(b)/len(b),1), len(b)) for a,b in acc[ph].items()})
PY
