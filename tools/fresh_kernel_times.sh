#!/bin/bash
# Development aid (GPU box): per-kernel times of the fresh_device mode from a rocprofv3 kernel trace (the first refill fills
# every ring and is listed apart from the periodic ones).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fdp
rocprofv3 --kernel-trace --output-format csv -d /tmp/fdp -- python3 $ROOT/tools/time_fresh_device.py 2>&1 | grep "us per step"
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("/tmp/fdp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    if len(v) < 3: continue
    w=sorted(v[1:])
    print("   %-60s calls %4d first %9.1f us | others: median %8.1f us sum %9.1f us" % (k, len(v), v[0], w[len(w)//2], sum(v[1:])))
PY
