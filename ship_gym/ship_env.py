"""`from ship_gym.ship_env import ShipEnv` (train/random.py:1, train/rllib/ppo.py:6)."""
from ship_sim_gym_amd.ship_env import DEFAULT_STATE_VAL, STEP_PENALTY, ShipEnv  # noqa: F401
from ship_sim_gym_amd.vec_env import ShipVecEnv  # noqa: F401
