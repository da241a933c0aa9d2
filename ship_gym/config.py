"""`from ship_gym.config import EnvConfig, GameConfig` (train/random.py:2, train/stable_baselines/ppo.py:14)."""
from ship_sim_gym_amd.config import *  # noqa: F401,F403
from ship_sim_gym_amd.config import EnvConfig, GameConfig, LidarConfig  # noqa: F401
