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
step_avg_ns = None
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    v = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    if v:
        step_avg_ns = sum(v) / len(v)
print("== PMC (mean per dispatch of the step kernel) ==")
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "step_kernel" not in row.get("Kernel_Name", ""):
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if row.get("End_Timestamp"):
            dur[row["Counter_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if "SQ_WAVE_CYCLES" in acc and "SQ_WAVES" in acc and dur["SQ_WAVE_CYCLES"]:
        # every wave of a launch is resident from its start to its end (one workgroup per CU, 4 waves per SIMD), SQ_WAVE_CYCLES
        # counts in units of 4 cycles: wave-cycles * 4 / waves = shader cycles of the launch; / its duration in THIS pass = the
        # shader clock the kernel actually ran at (the 2.4 GHz of the data sheet is the boost ceiling)
        big["shader_clock_ghz"] = (sum(acc["SQ_WAVE_CYCLES"]) * 4.0 / sum(acc["SQ_WAVES"])) / (sum(dur["SQ_WAVE_CYCLES"]) / len(dur["SQ_WAVE_CYCLES"]))
        print("shader clock during the launches of this pass: %.3f GHz" % big["shader_clock_ghz"])
    for k, v in acc.items():
        print("%-28s mean %.6g  (n=%d)   largest dispatch %.6g" % (k, sum(v) / len(v), len(v), max(v)))
        if k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
                 "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"):
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
    # what bench.py derives roofline.hbm_measured / valu_issue_frac / bound from (only when source_sha matches the tree's)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    per = float(envs * spl)
    cj = {"source_sha": bench.source_sha(), "mode": os.environ.get("SSG_PROF_MODE", "trajectory"),
          "source": os.environ.get("SSG_PROF_SOURCE", os.path.basename(out)), "envs": envs, "steps_per_launch": spl,
          "hbm_bytes_per_env_step": hbm / per, "fetch_bytes_per_env_step": big["FETCH_SIZE"] / fcal * 1024 / per,
          "write_bytes_per_env_step": big["WRITE_SIZE"] / wcal * 1024 / per,
          "avg_launch_us_profiled": (step_avg_ns or 0) / 1e3}
    if "SQ_INSTS_VALU" in big:
        cj["valu_wave_insts_per_env_step"] = big["SQ_INSTS_VALU"] / per
        cj["salu_wave_insts_per_env_step"] = big.get("SQ_INSTS_SALU", 0) / per
        cj["lds_wave_insts_per_env_step"] = big.get("SQ_INSTS_LDS", 0) / per
    if "SQ_LDS_IDX_ACTIVE" in big and step_avg_ns:
        # cycles the LDS pipe of a CU is busy / cycles of the launch (256 CUs, 2.4 GHz nominal)
        cj["lds_pipe_busy_frac"] = big["SQ_LDS_IDX_ACTIVE"] / 256.0 / (step_avg_ns * 2.4)
        cj["lds_active_cycles_per_env_step"] = big["SQ_LDS_IDX_ACTIVE"] / per  # summed over the 256 CUs' LDS pipes
        cj["lds_bank_conflict_frac_of_busy"] = big.get("SQ_LDS_BANK_CONFLICT", 0) / big["SQ_LDS_IDX_ACTIVE"]
    if "SQ_ACTIVE_INST_VALU" in big and "SQ_WAVE_CYCLES" in big:
        cj["valu_busy_frac_of_simd_cycles"] = big["SQ_ACTIVE_INST_VALU"] / (big["SQ_WAVE_CYCLES"] / 4.0)
        cj["wait_any_frac_of_wave_cycles"] = big.get("SQ_WAIT_ANY", 0) / big["SQ_WAVE_CYCLES"]
    if "shader_clock_ghz" in big:
        cj["shader_clock_ghz"] = big["shader_clock_ghz"]
    json.dump(cj, open(os.path.join(out, "counters.json"), "w"), indent=1)
    print("== counters.json ==")
    print(json.dumps(cj, indent=1))
