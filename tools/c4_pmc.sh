#!/bin/bash
# Development aid (GPU box): PMC passes over config 4 (tools/time_config4.py), per-kernel averages of each counter.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "${@}"; do
  i=$((i+1)); rm -rf /tmp/c4pmc_$i
  K=${K:-100} rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/c4pmc_$i -- python3 $ROOT/tools/time_config4.py > /tmp/c4pmc_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/c4pmc_$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k in acc:
    if "dyn" in k or "step_kernel" in k:
        print("  %-40s" % k, "  ".join("%s=%.4g" % (c, acc[k][c]/cnt[k][c]) for c in sorted(acc[k])))
PY
done
