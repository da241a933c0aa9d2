#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export K=60
timeout 150 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VALU --output-format csv -d /tmp/icp -- python3 $ROOT/tools/time_config4.py > /tmp/icp.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/icp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "dyn_step" in k or "step_kernel" in k or "dyn_sort" in k:
            acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {a: round(sum(b)/len(b),1) for a,b in v.items()})
PY
tail -3 /tmp/icp.log
