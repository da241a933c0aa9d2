#!/usr/bin/env python3
"""Timing of the per-episode-fresh-world mode (map_mode="fresh_device", ring R): run under rocprofv3 --kernel-trace --stats
to split a step's time between the gathering step kernel and the world generator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
R = int(os.environ.get("R", "32")); K = int(os.environ.get("K", "620"))  # (R=128 K=635: the bench form)
vec = ShipVecEnv(int(os.environ.get("N", "65536")), n_beams=8, map_mode="fresh_device", ring=R)
acts = vec.random_actions(12345, 0, K + 62)
vec.reset_tensor(); vec.rollout_tensor(acts[:62]); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); vec.rollout_tensor(acts[62:]); e1.record(); torch.cuda.synchronize()
print("fresh_device R=%d: %.2f us per step" % (R, e0.elapsed_time(e1) * 1e3 / K))
