#!/bin/bash
# Development aid (GPU box): per-kernel times of config 4 from rocprofv3 --kernel-trace --stats, for the product
# library or the variants named on the command line (SSG_LIB_PATH, tools/build_variant.sh).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for V in "${@:-product}"; do
  if [ "$V" = product ]; then unset SSG_LIB_PATH; else export SSG_LIB_PATH=$ROOT/ship_sim_gym_amd/libshipsim_$V.so; fi
  rm -rf /tmp/c4p_$V
  K=${K:-200} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4p_$V -- python3 $ROOT/tools/time_config4.py 2>&1 | grep us_per_step | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', 'profiled us_per_step %.1f' % d['us_per_step'])"
  python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/c4p_$V/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("   %-58s calls %5s avg %8.1f us" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3))
# the full step launch by launch (the first ones after the reset step every env; later bursts = episodes still in phase)
for f in glob.glob("/tmp/c4p_$V/**/*kernel_trace.csv", recursive=True):
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "dyn_step_kernel" in r["Kernel_Name"])
    if d:
        u = [x[1] / 1e3 for x in d]
        print("   dyn_step launches: median %.1f us; first 60:" % sorted(u)[len(u) // 2], " ".join("%.0f" % x for x in u[:60]))
PY
done
