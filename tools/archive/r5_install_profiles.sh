#!/bin/bash
# Copies the summaries of tools/r5_profiles.sh (gpurun_out/prof_r5h, prof_c4_r5) into profiles/r5 and installs the sha-tied counters.
cd "$(dirname "$0")/.."
mkdir -p profiles/r5/config4
P=gpurun_out/prof_r5h
cp $P/counters.json profiles/r5/counters.json; cp $P/counters.json profiles/counters_latest.json; cp $P/traffic.json profiles/r5/traffic.json
cp $P/summary.txt profiles/r5/rocprofv3_summary_bench_65536x8beam_trajectory.txt
cp "$(ls -t $(find $P/trace -name "*kernel_stats.csv") | head -1)" profiles/r5/kernel_stats.csv   # (gpurun_out accumulates earlier runs: the newest)
cp "$(ls -t $(find $P/trace -name "*domain_stats.csv") | head -1)" profiles/r5/domain_stats.csv
C=gpurun_out/prof_c4_r5
cp "$(ls -t $(find $C/trace -name "*kernel_stats.csv") | head -1)" profiles/r5/config4/kernel_stats.csv
cp "$(ls -t $(find $C/trace_memo_off -name "*kernel_stats.csv") | head -1)" profiles/r5/config4/kernel_stats_memo_off.csv
cp $C/kernel_times.txt profiles/r5/config4/kernel_times.txt
grep -A30 "== kernel stats ==" gpurun_out/prof_c4_r5.log > profiles/r5/config4/rocprofv3_summary_config4_memo.txt
python3 - <<'PY'
import ast
rows = {}
for l in open('gpurun_out/prof_c4_r5.log'):
    if "{'" not in l or not l.startswith("void ssg::"): continue
    name = l.split("{'")[0].strip()
    rows.setdefault(name, {}).update(ast.literal_eval("{'" + l.split("{'", 1)[1].strip()))
full = {"void ssg::dyn_step_kernel<true, true>(ss": "ssg::dyn_step_kernel<true, true>  (full cpSpaceStep of the queued envs, memo on)",
        "void ssg::step_kernel<10, 256, true, fal": "ssg::step_kernel<10, 256, true, false, true>  (the DYN step kernel, one step per launch)"}
out = ["# Config 4 (65 536 envs x 4 ships, 10 beams, bank mode, memo ON): rocprofv3 PMC passes over tools/time_config4.py (K = 300 + 50 warm-up steps),",
       "# each counter group in its own run with --kernel-trace only (tools/profile_c4.sh r5); MEAN per dispatch over the 350 launches of each kernel",
       "# (the first 8 launches after the full reset compute every env: ~170 us each).  FETCH_SIZE / WRITE_SIZE in KB as reported (FETCH_SIZE reads",
       "# half the bytes on gfx950: profiles/r5/rocprofv3_summary_bench_65536x8beam_trajectory.txt has the calibration).", ""]
for k, d in rows.items():
    out.append(full.get(k, k))
    for c in sorted(d): out.append("    %-24s %16.1f" % (c, d[c]))
    out.append("    -> VALU wave-instructions per env-step: %.2f; HBM bytes per env-step: fetch %.1f (x2 calibration) + write %.1f; SQ busy %.0f cycles per dispatch" % (
        d["SQ_INSTS_VALU"] / 65536.0, 2 * d.get("FETCH_SIZE", 0) * 1024 / 65536.0, d.get("WRITE_SIZE", 0) * 1024 / 65536.0, d["SQ_BUSY_CYCLES"]))
    out.append("")
open('profiles/r5/config4/pmc_summary.txt', 'w').write("\n".join(out))
import json, sys
sys.path.insert(0, '.')
import bench
print("tree sha", bench.source_sha(), "profile sha", json.load(open('profiles/r5/counters.json'))['source_sha'])
PY
