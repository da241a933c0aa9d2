#!/usr/bin/env python3
"""Development aid: wall-clock of the driver-style timed region (one 20-step launch bracketed by synchronize) under
different completion-wait strategies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
import bench
K, R = 20, 15
vec = ShipVecEnv(65536, n_maps=64, n_beams=8)
acts = vec.random_actions(1, 0, 5 + K * R * 4)
vec.reset_tensor()
bufs = bench.traj_buffers(vec, K, 5)
vec.rollout_tensor(acts[:5], trajectory=True, out=bufs[0])
off = [5]
def run(mode):
    walls, evs = [], []
    for r in range(R):
        a = acts[off[0]: off[0] + K]; off[0] += K
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        vec.rollout_tensor(a, trajectory=True, out=bufs[r % 5])
        e1.record()
        if mode == "spin":
            while not e1.query():
                pass
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e6)
        evs.append(e0.elapsed_time(e1) * 1e3)
    walls.sort(); evs.sort()
    print("%-6s HSA_ENABLE_INTERRUPT=%s: wall median %.1f us (min %.1f), events median %.1f us (min %.1f); first repeats wall %s" % (
        mode, os.environ.get("HSA_ENABLE_INTERRUPT", "unset"), walls[R // 2], walls[0], evs[R // 2], evs[0], ""))
run("sync"); run("spin"); run("sync")
