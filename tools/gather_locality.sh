#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/glp
rocprofv3 --kernel-trace --output-format csv -d /tmp/glp -- python3 $ROOT/tools/gather_locality_probe.py 2>&1 | grep "us per step"
python3 - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob("/tmp/glp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"][:70], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
rows.sort()
# step-kernel launches in time order, grouped into runs separated by other big kernels
cur=None; out=[]
for t,k,d in rows:
    if "step_kernel" in k:
        out.append(d)
print("step_kernel launches: %d" % len(out))
import statistics
# print consecutive groups of launch durations (us)
print(" ".join("%.0f" % d for d in out))
PY
