#!/bin/bash
# round 4: kernel trace of config 4 (pipelined and serial schedules), stats only.  usage: r4_prof_c4.sh <outdir> [variants...]
# variant = name:ENV=V,ENV=V  e.g.  pipe:  serial:SSG_DYN_SERIAL=1  b64g:SSG_BLOCK=64,BIG=1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r4d}
shift
VARS="$@"
[ -z "$VARS" ] && VARS="pipe: serial:SSG_DYN_SERIAL=1"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export K=${K:-100}
for var in $VARS; do
  name=${var%%:*}; envs=${var#*:}
  ( for kv in ${envs//,/ }; do export $kv; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 $ROOT/tools/time_config4.py > $OUT/trace_$name.log 2>&1 )
  echo "== $name ($envs): $(grep -o '"us_per_step": [0-9.]*' $OUT/trace_$name.log)"
  python3 - "$OUT/trace_$name" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print("   %-60s calls %6s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  find $OUT/trace_$name -name "*kernel_trace.csv" -size +3M -delete
done
