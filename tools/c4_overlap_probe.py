#!/usr/bin/env python3
"""Do config 4's two kernels share the chip?  Two independent handles (65 536 envs x 4 ships each) stepped on two HIP
streams at once: if the full dyn step of one (a few hundred lone waves, chain-bound, ~10 % of the VALU issue rate) and the
step kernel of the other (256 workgroups of 1 024 threads, a whole CU's LDS each) overlapped perfectly, both rollouts
together would take the time of one.  What the pipelined config-4 rollout (dyn step k+1 beside step kernel k) can hope for."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv

n = int(os.environ.get("N", "65536"))
K, W = int(os.environ.get("K", "200")), 50
prio = os.environ.get("PRIO", "0") == "1"
dev = torch.device("cuda:0")
vecs = [ShipVecEnv(n, n_beams=10, n_maps=64, n_ships=4) for _ in range(2)]
acts = [v.random_actions(12345 + i, 0, K + W) for i, v in enumerate(vecs)]
streams = [torch.cuda.Stream(dev, priority=(-1 if (prio and i == 0) else 0)) for i in range(2)]
for v, a in zip(vecs, acts):
    v.reset_tensor()
    v.rollout_tensor(a[:W])
torch.cuda.synchronize()


def timed(which):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in which]
    torch.cuda.synchronize()
    for (e0, e1), i in zip(ev, which):
        with torch.cuda.stream(streams[i]):
            e0.record()
            vecs[i].rollout_tensor(acts[i][W:])
            e1.record()
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1) * 1e3 / K for e0, e1 in ev]


for rep in range(2):
    a = timed([0])[0]
    b = timed([1])[0]
    both = timed([0, 1])
    # the second rollout is launched after the first one's K x 3 launches were queued: offset its phase
    print("alone %.1f / %.1f us per step; together %.1f / %.1f us per step (serial would be %.1f)" % (a, b, both[0], both[1], a + b))
