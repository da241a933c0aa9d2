#!/usr/bin/env python3
"""Development aid: is a single-step launch of the step kernel slower when another kernel has swept the caches in between?
(Config 4's step kernel runs after the dyn kernels every step; run under rocprofv3 --kernel-trace --stats and compare the
step kernel's average duration with THRASH=0 / THRASH=<MB>.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
n = int(os.environ.get("SSG_N", "65536")); nb = int(os.environ.get("SSG_NB", "10"))
mb = int(os.environ.get("THRASH", "0"))
vec = ShipVecEnv(n, n_maps=64, n_beams=nb)
junk = torch.zeros(max(mb, 1) * (1 << 20) // 8, dtype=torch.float64, device="cuda")
acts = vec.random_actions(12345, 0, 700)
vec.reset_tensor()
for k in range(100): vec.step_tensor(acts[k])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(100, 700):
    vec.step_tensor(acts[k])
    if mb: junk.add_(1.0)
e1.record(); torch.cuda.synchronize()
print("THRASH=%d MB: %.2f us per step (events, incl. the sweep)" % (mb, e0.elapsed_time(e1) * 1e3 / 600))
