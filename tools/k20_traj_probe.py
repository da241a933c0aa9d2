#!/usr/bin/env python3
"""Development aid: where do a short trajectory-mode launch's extra microseconds come from?  HIP-event time of one K-step
launch (65 536 envs, 8 beams) for overwrite vs trajectory outputs, with one reused / several rotating / pre-touched buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
import bench

K = int(os.environ.get("K", "20"))
R = 12
vec = ShipVecEnv(65536, n_maps=64, n_beams=8)
acts = vec.random_actions(1, 0, 200 + K * R * 6)
vec.reset_tensor()
vec.rollout_tensor(acts[:200])


def timeit(fn, reps=R):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); fn(r); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


off = [200]
def nxt():
    a = acts[off[0]: off[0] + K]; off[0] += K; return a

print("K = %d" % K)
print("overwrite            : median %.1f us (min %.1f max %.1f)" % timeit(lambda r: vec.rollout_tensor(nxt())))
one = bench.traj_buffers(vec, K, 1)[0]
print("trajectory, 1 buffer : median %.1f us (min %.1f max %.1f)" % timeit(lambda r: vec.rollout_tensor(nxt(), trajectory=True, out=one)))
many = bench.traj_buffers(vec, K, R)
print("trajectory, %2d fresh : median %.1f us (min %.1f max %.1f)" % ((R,) + timeit(lambda r: vec.rollout_tensor(nxt(), trajectory=True, out=many[r]))))
for b in many:
    for t in b:
        t.zero_()
torch.cuda.synchronize()
print("trajectory, %2d touched: median %.1f us (min %.1f max %.1f)" % ((R,) + timeit(lambda r: vec.rollout_tensor(nxt(), trajectory=True, out=many[r]))))
