"""Import-compatibility alias: `ship_gym.*` names of the reference resolve to the MI355X-native implementation.

With this repository on PYTHONPATH the reference's caller scripts (train/random.py, train/stable_baselines/ppo.py,
train/rllib/ppo.py, train/rllib/pbt.py) import `ship_gym.ship_env.ShipEnv` / `ship_gym.config.*` unchanged and get
the HIP-backed env.  Nothing here is reference code; every module re-exports from ship_sim_gym_amd."""
