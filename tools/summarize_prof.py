#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory: per-kernel time stats and mean PMC counters per dispatch."""
import csv, glob, os, sys, collections, statistics

out = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")})
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    d = collections.defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        d[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[n] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for n, v in d.items():
        v2 = sorted(v)
        print("trace:", n[:60], "calls", len(v), "avg_us %.3f med_us %.3f min_us %.3f" % (sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3), meta[n])
print("== PMC (mean per dispatch of the step kernel) ==")
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "step_kernel" not in row.get("Kernel_Name", ""):
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s mean %.6g  (n=%d)" % (k, sum(v) / len(v), len(v)))
