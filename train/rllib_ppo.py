#!/usr/bin/env python3
"""RLlib PPO on the MI355X batched env: the counterpart of the reference's train/rllib/ppo.py (:10-44) with ONE thing
changed — `env_creator` returns a `ShipVecEnv(..., rllib=True)` (an RLlib VectorEnv: vector_step reports terminal
observations, reset_at is the one reset) instead of a single `ShipEnv`, so one rollout worker steps `--envs` envs on the
GPU where the reference spreads one env per worker over `cpu_count() - 1` processes.  Same game configuration (FPS 100000,
SPEED 40, DEBUG on, BOUNDS 1000x1000), same experiment: PPO, 12 h, num_sgd_iter 10, minibatch 2048, train batch 10000,
the three-point learning-rate schedule.

ray (0.6.0 in the reference's requirements) is not part of this image: imports are guarded;
`tests/test_trainer_scripts.py` drives main() through env creation and the first vector_reset / vector_step under stand-ins."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ship_gym.config import EnvConfig, GameConfig  # noqa: E402

ENV_NAME = "ship-gym-v1"


def game_configuration(speed=40, fps=100000, debug=True):
    gc = GameConfig
    gc.FPS = fps
    gc.SPEED = speed
    gc.DEBUG = debug
    gc.BOUNDS = (1000, 1000)
    return gc


def make_env_creator(num_envs, device, game_config):
    def env_creator(_env_config):
        """THE changed line: a VectorEnv of num_envs envs instead of `ShipEnv(game_config, env_config)`."""
        from ship_sim_gym_amd.vec_env import ShipVecEnv
        return ShipVecEnv(num_envs, game_config, EnvConfig, device=device, rllib=True)
    return env_creator


def experiment(num_workers):
    return {
        "shipgym_best": {
            "run": "PPO",
            "stop": {"time_total_s": 12 * 60 * 60},
            "env": ENV_NAME,
            "config": {
                "num_gpus": 1,
                "num_workers": num_workers,
                "num_sgd_iter": 10,
                "sgd_minibatch_size": 2048,
                "train_batch_size": 10000,
                "lr_schedule": [[0, 0.001], [5e6, 0.0001], [1e7, 0.00001]],
            },
        },
    }


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096, help="envs per rollout worker (the batch on one GPU)")
    ap.add_argument("--workers", type=int, default=1, help="rollout workers (the reference: cpu_count() - 1, one env each)")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    try:
        import ray
        from ray import tune
    except ImportError as e:
        sys.exit("ray is not installed (%s): pip install ray==0.6.0" % e)
    ray.init(num_gpus=1)
    tune.register_env(ENV_NAME, make_env_creator(args.envs, args.device, game_configuration()))
    tune.run_experiments(experiment(args.workers))


if __name__ == "__main__":
    main()
