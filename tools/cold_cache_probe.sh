cd /tmp && export TMPDIR=/tmp
for T in 0 64 512; do
rm -rf /tmp/ccp; THRASH=$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ccp -- python3 $GRAFT_REPO_ROOT/tools/cold_cache_probe.py 2>&1 | grep THRASH
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/ccp/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print("   %-58s calls %5s avg %8.1f us" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
