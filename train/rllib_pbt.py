#!/usr/bin/env python3
"""RLlib PPO under population-based training on the MI355X batched env: the counterpart of the reference's
train/rllib/pbt.py (:14-72) with the env-construction line changed as in train/rllib_ppo.py (a `ShipVecEnv(..., rllib=True)`
per rollout worker instead of one `ShipEnv` per worker).  Same game configuration (FPS 1000, SPEED 30, BOUNDS 1000x1000),
same scheduler (perturb every 600 s of training time on episode_reward_mean, resample 0.33, the six mutated
hyper-parameters with the reference's ranges), same experiment (120 samples, kl_coeff 1.0, lambda 0.95, clip 0.2, lr 5e-4,
randomly drawn num_sgd_iter / minibatch / train batch).  ray is not part of this image: imports are guarded."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from train.rllib_ppo import game_configuration, make_env_creator  # noqa: E402

ENV_NAME = "ShipGym-v1"


def mutations():
    return {
        "lambda": lambda: random.uniform(0.9, 1.0),
        "clip_param": lambda: random.uniform(0.01, 0.5),
        "lr": [1e-3, 5e-4, 1e-4, 5e-5, 1e-5],
        "num_sgd_iter": lambda: random.randint(1, 30),
        "sgd_minibatch_size": lambda: random.randint(128, 16384),
        "train_batch_size": lambda: random.randint(2000, 160000),
    }


def experiment(num_workers):
    return {
        "pbt_ship_sim_v2": {
            "run": "PPO",
            "env": ENV_NAME,
            "num_samples": 120,
            "checkpoint_at_end": True,
            "checkpoint_freq": 2,
            "config": {
                "kl_coeff": 1.0,
                "num_workers": num_workers,
                "num_gpus": 1,
                "lambda": 0.95,
                "clip_param": 0.2,
                "lr": 5.0e-4,
                "num_sgd_iter": lambda spec: random.choice([10, 20, 30]),
                "sgd_minibatch_size": lambda spec: random.choice([128, 512, 2048]),
                "train_batch_size": lambda spec: random.choice([10000, 20000, 40000]),
            },
        },
    }


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    try:
        import ray
        from ray.tune import register_env, run_experiments
        from ray.tune.schedulers import PopulationBasedTraining
    except ImportError as e:
        sys.exit("ray is not installed (%s): pip install ray==0.6.0" % e)
    register_env(ENV_NAME, make_env_creator(args.envs, args.device, game_configuration(speed=30, fps=1000, debug=False)))
    pbt = PopulationBasedTraining(time_attr="time_total_s", reward_attr="episode_reward_mean", perturbation_interval=600,
                                  resample_probability=0.33, hyperparam_mutations=mutations())
    ray.init()
    run_experiments(experiment(args.workers), scheduler=pbt)


if __name__ == "__main__":
    main()
