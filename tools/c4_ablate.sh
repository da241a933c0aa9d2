#!/bin/bash
# Development aid (GPU box; needs `tools/build_variant.sh abl -DSSG_ABLATION`): config 4's per-kernel times with sections of
# the step kernel switched off (timing only: the outputs are wrong; the dyn kernels' work does not depend on them).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export SSG_LIB_PATH=$ROOT/ship_sim_gym_amd/libshipsim_abl.so
for A in ${@:-0 0x80}; do
  rm -rf /tmp/c4abl
  SSG_ABLATE=$A rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4abl -- python3 $ROOT/bench.py --workload c4 --steps 300 --warmup 50 --no-cpu-baseline > /tmp/c4abl.log 2>&1
  echo "ablate=$A"
  python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/c4abl/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("   %-58s calls %5s avg %8.1f us" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
