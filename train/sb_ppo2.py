#!/usr/bin/env python3
"""stable-baselines PPO2 on the MI355X batched env: the counterpart of the reference's train/stable_baselines/ppo.py
(make_env :54-76, train :84-100, main :102-143) with ONE thing changed — where the reference builds
`SubprocVecEnv([make_env() for i in range(num_cpu)])` (one OS process per env, :122-123), this script builds one
`ShipVecEnv` of `--envs` envs on the GPU.  Same game configuration (FPS 1000, SPEED 30, DEBUG on, BOUNDS 1000x1000),
same three linearly decaying learning rates (1e-3, 1e-4, 1e-5 -> 0), same 1e6 timesteps each, same MlpPolicy, same
TensorBoard / model directories.

stable-baselines (2.2.0 in the reference's requirements) is not part of this image: the imports are guarded, and
`tests/test_trainer_scripts.py` drives main() up to the first reset / step_async / step_wait under stand-in modules."""
import argparse
import os
import sys
import time
from datetime import datetime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ship_gym.config import EnvConfig, GameConfig  # noqa: E402  (the reference's import line; the alias package)

LOG_DIR, MODEL_DIR = "logs/learning", "models"


def game_configuration():
    """train/stable_baselines/ppo.py:65-69"""
    gc = GameConfig
    gc.FPS = 1000
    gc.SPEED = 30
    gc.DEBUG = True
    gc.BOUNDS = (1000, 1000)
    return gc


def make_vec_env(num_envs, device="cuda:0", **kw):
    """THE changed line: a batched env instead of SubprocVecEnv over per-process ShipEnvs."""
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    return ShipVecEnv(num_envs, game_configuration(), EnvConfig, device=device, **kw)


def decaying(start, stop=0.0):
    """lr(frac) with frac = remaining progress in [1, 0] (train/stable_baselines/ppo.py:113-118)"""
    return lambda frac: start + (stop - start) * (1 - frac)


def train(model_cls, policy, tid, env, lr, steps, tb_root):
    t0 = time.time()
    model = model_cls(policy, env, learning_rate=lr, verbose=1, tensorboard_log=os.path.join(tb_root, "%d_%s" % (tid, model_cls.__name__)))
    model.learn(total_timesteps=steps, log_interval=10000)
    dt = time.time() - t0
    print("Trained %d steps in %.1f s = %.0f steps/min" % (steps, dt, steps / max(dt, 1e-9) * 60))
    model.save(os.path.join(MODEL_DIR, "result_lr%s" % tid))
    return model


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096, help="envs in the batch (the reference: one per CPU core)")
    ap.add_argument("--steps", type=int, default=int(1e6))
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    try:
        from stable_baselines import PPO2
        from stable_baselines.common.policies import MlpPolicy
    except ImportError as e:
        sys.exit("stable-baselines is not installed (%s): pip install stable-baselines==2.2.0, or see train/ppo_torch.py for a "
                 "plain-PyTorch PPO loop over the same env" % e)
    os.makedirs(LOG_DIR, exist_ok=True)
    os.makedirs(MODEL_DIR, exist_ok=True)
    tb_root = os.path.join(LOG_DIR, "tb", str(int(time.time())))
    env = make_vec_env(args.envs, args.device)
    for tid, lr0 in enumerate((1.0e-3, 1.0e-4, 1.0e-5), start=1):
        print("Started training at %s: %d steps, learning rate %g -> 0" % (datetime.now(), args.steps, lr0))
        train(PPO2, MlpPolicy, tid, env, decaying(lr0), args.steps, tb_root)
    env.close()


if __name__ == "__main__":
    main()
