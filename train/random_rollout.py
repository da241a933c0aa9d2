#!/usr/bin/env python3
"""The reference's train/random.py (random-action rollout, :9-27) against the batched env: N envs, K steps,
actions from the device-side Philox stream, 100 steps fused per launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ship_gym.config import EnvConfig, GameConfig  # noqa: E402
from ship_sim_gym_amd.vec_env import ShipVecEnv  # noqa: E402

if __name__ == "__main__":
    n, K = int(os.environ.get("ENVS", "65536")), int(os.environ.get("STEPS", "1000"))
    env = ShipVecEnv(n, GameConfig, EnvConfig)
    env.reset_tensor()
    acts = env.random_actions(seed=0, step0=0, K=K)
    # TRAJ=1: keep what the reference's loop sees at every step — `ret = env.step(...)`, train/random.py:20 — as [K, N, ...]
    # tensors in HBM (ssg_rollout_traj) instead of only the last step's outputs
    traj = os.environ.get("TRAJ", "0") != "0"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = env.rollout_tensor(acts, trajectory=traj)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if traj:
        print("trajectory tensors: obs %s, reward %s, done %s; mean step reward %+.4f" % (
            tuple(out[0].shape), tuple(out[1].shape), tuple(out[2].shape), float(out[1].mean())))
    st = env.stats()
    print("%d envs x %d steps in %.3f s = %.2f G env-steps/s; %d episodes, mean return %+.3f, %.2f goals per episode" % (
        n, K, dt, n * K / dt / 1e9, st["episodes"], st["sum_return"] / max(st["episodes"], 1),
        st["goals_hit"] / max(st["episodes"], 1)))
