#!/usr/bin/env python3
"""Diagnostic: 20-step launches back to back vs separated by a synchronize (run under rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
vec = ShipVecEnv(65536, n_maps=64, n_beams=8)
acts = vec.random_actions(12345, 0, 1000)
vec.reset_tensor(); vec.rollout_tensor(acts[:200]); torch.cuda.synchronize()
def timed(label, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    print("%-50s events %.1f us wall %.1f us" % (label, e0.elapsed_time(e1) * 1e3, (time.perf_counter() - t0) * 1e6))
for r in range(3):
    timed("one 20-step launch after sync", lambda: vec.rollout_tensor(acts[200:220]))
timed("10 x 20-step launches back to back", lambda: [vec.rollout_tensor(acts[200 + 20 * i: 220 + 20 * i]) for i in range(10)])
for gap_us in (0, 100, 1000, 10000):
    time.sleep(gap_us * 1e-6)
    timed("one 20-step launch after %d us of extra idle" % gap_us, lambda: vec.rollout_tensor(acts[200:220]))
timed("one 100-step launch after sync", lambda: vec.rollout_tensor(acts[200:300]))
timed("one 200-step rollout (2 launches)", lambda: vec.rollout_tensor(acts[200:400]))
