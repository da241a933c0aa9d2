#!/bin/bash
# Copies the summaries tools/final_profiles.sh <tag> left under gpurun_out/ into profiles/<tag>/ and installs the sha-tied counters
# (profiles/counters_latest.json: what bench.py derives roofline.traffic / valu_issue_frac from).   tools/install_profiles.sh r6
R=${1:-r6}
cd "$(dirname "$0")/.."
mkdir -p profiles/$R/config4
P=gpurun_out/prof_${R}h
cp $P/counters.json profiles/$R/counters.json; cp $P/counters.json profiles/counters_latest.json; cp $P/traffic.json profiles/$R/traffic.json
cp $P/summary.txt profiles/$R/rocprofv3_summary_bench_65536x8beam_trajectory.txt
cp "$(ls -t $(find $P/trace -name "*kernel_stats.csv") | head -1)" profiles/$R/kernel_stats.csv
cp "$(ls -t $(find $P/trace -name "*domain_stats.csv") | head -1)" profiles/$R/domain_stats.csv
C=gpurun_out/prof_c4_$R
cp "$(ls -t $(find $C/trace -name "*kernel_stats.csv") | head -1)" profiles/$R/config4/kernel_stats.csv
cp "$(ls -t $(find $C/trace_memo_off -name "*kernel_stats.csv") | head -1)" profiles/$R/config4/kernel_stats_memo_off.csv
cp $C/kernel_times.txt profiles/$R/config4/kernel_times.txt
grep -A30 "== kernel stats ==" gpurun_out/prof_c4_$R.log > profiles/$R/config4/rocprofv3_summary_config4_memo.txt
F=gpurun_out/${R}final
cp $F/bench_line.json profiles/$R/bench_line.json
for i in 1 2 3; do cp $F/bench_line_driver_form_$i.json profiles/$R/bench_line_driver_form_$i.json; done
cp $F/soak_parity.txt profiles/$R/soak_parity.txt
R=$R python3 - <<'PY'
import ast, json, os, sys
R = os.environ["R"]
rows = {}
for l in open('gpurun_out/prof_c4_%s.log' % R):
    if "{'" not in l or not l.startswith("void ssg::"): continue
    name = l.split("{'")[0].strip()
    rows.setdefault(name, {}).update(ast.literal_eval("{'" + l.split("{'", 1)[1].strip()))
full = {"void ssg::dyn_step_kernel<true, true>(ss": "ssg::dyn_step_kernel<true, true>  (full cpSpaceStep of the queued envs, memo on)",
        "void ssg::step_kernel<10, 256, true, fal": "ssg::step_kernel<10, 256, true, false, true>  (the DYN step kernel, one step per launch)"}
out = ["# Config 4 (65 536 envs x 4 ships, 10 beams, bank mode, memo ON): rocprofv3 PMC passes over tools/time_config4.py (K = 300 + 50 warm-up steps),",
       "# each counter group in its own run with --kernel-trace only (tools/profile_c4.sh); MEAN per dispatch over the 350 launches of each kernel",
       "# (the first 8 launches after the full reset compute every env).  FETCH_SIZE / WRITE_SIZE in KB as reported (FETCH_SIZE reads",
       "# half the bytes on gfx950: the headline summary in this directory has the calibration).", ""]
for k, d in rows.items():
    out.append(full.get(k, k))
    for c in sorted(d): out.append("    %-24s %16.1f" % (c, d[c]))
    if "SQ_INSTS_VALU" in d:
        out.append("    -> VALU wave-instructions per env-step: %.2f; HBM bytes per env-step: fetch %.1f (x2 calibration) + write %.1f; SQ busy %.0f cycles per dispatch" % (
            d["SQ_INSTS_VALU"] / 65536.0, 2 * d.get("FETCH_SIZE", 0) * 1024 / 65536.0, d.get("WRITE_SIZE", 0) * 1024 / 65536.0, d.get("SQ_BUSY_CYCLES", 0)))
    out.append("")
open('profiles/%s/config4/pmc_summary.txt' % R, 'w').write("\n".join(out))
sys.path.insert(0, '.')
import bench
print("tree sha", bench.source_sha(), "profile sha", json.load(open('profiles/%s/counters.json' % R))['source_sha'])
PY
