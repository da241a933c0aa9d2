#!/usr/bin/env python3
"""Why do the first timed repeats of a driver-form bench run (20-step launches into DISTINCT trajectory buffer sets) take longer than
the later ones?  Event times of 20-step launches under different buffer histories.  GPU box: python3 tools/coldbuf_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
import bench

dev = torch.device("cuda", 0)
vec = ShipVecEnv(65536, device=dev, map_mode="bank", n_maps=64, map_seed=1000, n_beams=8)
K = 20
acts = vec.random_actions(12345, 0, K * 40)
scratch = bench.traj_buffers(vec, K, 1)[0]

def cond(ms=300):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8): vec.rollout_tensor(acts[:K], trajectory=True, out=scratch)
        torch.cuda.synchronize()

def timed(sets, order, label, pre=None):
    ts = []
    for i, si in enumerate(order):
        if pre: pre(sets[si])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(); vec.rollout_tensor(acts[i * K:(i + 1) * K], trajectory=True, out=sets[si]); e1.record()
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) * 1e6
        ts.append("%.0f/%.0f" % (e0.elapsed_time(e1) * 1e3, w))
    print(label, " ".join(ts), flush=True)

vec.reset_tensor()
sets = bench.traj_buffers(vec, K, 10)
for s_ in sets:
    for t in s_: t.zero_()
cond()
timed(sets, list(range(10)), "A  10 distinct sets zeroed BEFORE the conditioning (event/wall us):")
timed(sets, list(range(10)), "A2 the same 10 sets again:")
cond()
timed(sets, [0] * 10, "B  one set ten times after conditioning:")
cond()
for s_ in sets:
    for t in s_: t.zero_()
timed(sets, list(range(10)), "C  10 sets zeroed right before (after conditioning):")
cond()
timed(sets, list(range(10)), "D  10 sets, conditioning in between, not re-zeroed:")
sets2 = bench.traj_buffers(vec, K, 10)
cond()
timed(sets2, list(range(10)), "E  10 brand-new sets never touched:")
timed(sets2, list(range(10)), "E2 again:")
def touch(s_):
    for t in s_: t[:1].zero_()
cond()
timed(sets, list(range(10)), "F  each set's first rows zeroed right before its launch:", pre=touch)
