#!/bin/bash
# Development aid: register / scratch usage of the BASELINE instantiations of the step kernel (no GPU needed).
cd "$(dirname "$0")/../ship_sim_gym_amd/csrc" || exit 1
for G in 1 2; do
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. -DSSG_NB_GROUP=$G \
  -c shipsim_kernels.hip -o /tmp/_spill_check.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None; rec={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); rec[name]={}
    for key in ('VGPRs','ScratchSize \[bytes/lane\]','VGPRs Spill','SGPRs Spill'):
        m=re.search(key+r': (\d+)',l)
        if m and name: rec[name][key]=int(m.group(1))
    if 'error' in l: print(l.strip())
for n,r in rec.items():
    m=re.search(r'step_kernelILi(\d+)ELi(\d+)ELb([01])ELb([01])',n)
    if not m: continue
    nb,epw,lds,ex=map(int,m.groups())
    if (nb in (8,10)) and ex==0 and (epw==256 or (nb==10 and epw==128)) and lds==1:
        print('NB=%d EPW=%d LDS=%d EXACT=%d :'%(nb,epw,lds,ex), r)
"
done
