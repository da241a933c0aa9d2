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
big = {}
print("== PMC (mean per dispatch of the step kernel) ==")
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "step_kernel" not in row.get("Kernel_Name", ""):
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s mean %.6g  (n=%d)   largest dispatch %.6g" % (k, sum(v) / len(v), len(v), max(v)))
        if k in ("FETCH_SIZE", "WRITE_SIZE"):
            big[k] = sum(v) / len(v)  # every rollout launch runs the same 100 steps

print("== FETCH_SIZE / WRITE_SIZE calibration (tools/calib_pmc.py: 1 GiB read + 1 GiB written per dispatch, 8 B per lane) ==")
cal = {}
for name in ("cal_fetch", "cal_write"):
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "calib_copy8" in r.get("Kernel_Name", "")]
        if v:
            cal[name] = sum(v) / len(v)
            print("%s mean per dispatch: %.6g (KB units -> %.4f of the true 1 GiB)" % (name, cal[name], cal[name] * 1024 / 2**30))

if "FETCH_SIZE" in big and "WRITE_SIZE" in big:
    fcal = cal.get("cal_fetch", 524288.0) * 1024 / 2**30
    wcal = cal.get("cal_write", 1048576.0) * 1024 / 2**30
    hbm = (big["FETCH_SIZE"] / fcal + big["WRITE_SIZE"] / wcal) * 1024
    print("== HBM traffic of one 100-step rollout launch: (FETCH_SIZE/%.3f + WRITE_SIZE/%.3f) KB = %.6g bytes per launch ==" % (fcal, wcal, hbm))
    import json
    envs = int(os.environ.get("SSG_PROF_ENVS", "65536"))
    spl = int(os.environ.get("SSG_FUSE", "100"))
    json.dump({"hbm_bytes_per_launch": hbm, "hbm_bytes_per_env_step": hbm / (envs * spl), "envs": envs,
               "steps_per_launch": spl, "fetch_size_kb": big["FETCH_SIZE"], "write_size_kb": big["WRITE_SIZE"],
               "fetch_calibration": fcal, "write_calibration": wcal,
               "source": os.environ.get("SSG_PROF_SOURCE", os.path.basename(out))},
              open(os.path.join(out, "traffic.json"), "w"))
