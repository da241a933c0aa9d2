#!/bin/bash
# Development aid (GPU box; needs tools/build_variant.sh abl -DSSG_ABLATION): VALU / SALU / LDS wave-instructions per
# tile-step of the headline kernel with one timing-only ablation bit set at a time (what each component costs).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export SSG_LIB_PATH=$ROOT/ship_sim_gym_amd/libshipsim_abl.so
for A in ${ABLS:-0 1 2 8 0x10 0x20 0x40 0x80}; do
  rm -rf /tmp/ablpmc
  SSG_ABLATE=$A rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/ablpmc -- python3 $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-single-step --no-other-configs > /tmp/ablpmc.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/ablpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in acc.items()}
ts=1024*100.0
print("ablate %-5s VALU %6.0f SALU %6.0f LDS %5.0f per tile-step | VALU busy %.2f" % ("$A", m["SQ_INSTS_VALU"]/ts, m["SQ_INSTS_SALU"]/ts, m["SQ_INSTS_LDS"]/ts, m["SQ_ACTIVE_INST_VALU"]/(m["SQ_WAVE_CYCLES"]/4)))
PY
done
