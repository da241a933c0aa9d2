#!/usr/bin/env python3
"""Timing of BASELINE configs[3] (65 536 envs x 4 ships, 10 beams): dyn kernel + DYN step kernel per step."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv

n = int(os.environ.get("N", "65536"))
K, W = int(os.environ.get("K", "300")), 50
vec = ShipVecEnv(n, n_beams=10, n_maps=int(os.environ.get("MAPS", "64")), n_ships=4, dyn_memo=os.environ.get("MEMO", "1") != "0",
                 map_mode=os.environ.get("MAP_MODE", "bank"), ring=int(os.environ.get("RING", "8")))
acts = vec.random_actions(12345, 0, K + W)
vec.reset_tensor()
vec.rollout_tensor(acts[:W])
torch.cuda.synchronize()
memo0 = vec.dyn_memo_stats()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
vec.rollout_tensor(acts[W:])
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
B = 96 * 4 + 8 + 8 + 2 + 80 + 4 + 4 + 16 * 10 + 8 * 16 + 8 * 2 * 16 + 9   # SURVEY 8(d), S=4, nb=10, H=2 = 1043
from ship_sim_gym_amd import _native as N
fl = vec.field(N.F_DYN_FLAGS)
rest_frac = float(((fl & 4) != 0).double().mean())
steps = vec.field(N.F_STEP_COUNT)
print("rest fraction %.3f; mean step_count %.1f; rest among step_count>=6: %.3f" % (
    rest_frac, float(steps.double().mean()), float(((fl & 4) != 0)[steps >= 6].double().mean())))
print(json.dumps({"config": "C4 %d envs x 4 ships, 10 beams" % n, "us_per_step": ms * 1e3 / K,
                  "env_steps_per_s": n * K / (ms * 1e-3), "algorithmic_GBps": B * n * K / (ms * 1e-3) / 1e9,
                  "stats": vec.stats(), "memo": vec.dyn_memo_stats(), "memo_after_warmup": memo0, "memo_on": os.environ.get("MEMO", "1") != "0"}))
