#!/usr/bin/env python3
"""How the time of a 20-step launch depends on how long ago the batch was reset (all envs start their first episode in the same
step, so the first few hundred steps after a full reset are phase-locked): HIP-event time of consecutive 20-step trajectory launches
after reset_tensor(), and the same after a long burn-in.  GPU box: python3 tools/phase_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
import bench

dev = torch.device("cuda", 0)
vec = ShipVecEnv(65536, device=dev, map_mode="bank", n_maps=64, map_seed=1000, n_beams=8)
K = 20
acts = vec.random_actions(12345, 0, K * 64)
out = bench.traj_buffers(vec, K, 1)[0]
for t in out: t.zero_()
# condition the device on this env first
vec.reset_tensor()
for i in range(400): vec.rollout_tensor(acts[:K], trajectory=True, out=out)
torch.cuda.synchronize()

def run(label, n):
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); vec.rollout_tensor(acts[i * K:(i + 1) * K], trajectory=True, out=out); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(label, " ".join("%.0f" % t for t in ts))

vec.reset_tensor()
run("after a full reset, launches of 20 steps (us):", 40)
run("steady state (800 steps later):", 10)
st = vec.stats()
print("episodes", st["episodes"], "mean length", st["sum_length"] / max(1, st["episodes"]))
