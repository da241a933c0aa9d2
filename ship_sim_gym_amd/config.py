"""Configuration bags with the reference's names and defaults (ship_gym/config.py:8-24).

Like the reference, scripts mutate these class attributes in place (train/stable_baselines/ppo.py:65-69,
train/rllib/ppo.py:12-16) and the env snapshots them at construction time (game.py:37-45, ship_env.py:30,44-47).
``LidarConfig`` is accepted but — exactly as in the reference, where LiDAR hard-codes its own defaults
(models.py:29,150) — it is NOT read unless ``EnvConfig.USE_LIDAR_CONFIG`` is set (an extension; BASELINE config 3
needs 8 beams, which the reference cannot express).
"""


class LidarConfig(object):
    ANGULAR_SPREAD = 180  # degrees; dead in the reference (models.py:29 uses spread=90)
    DISTANCE = 100
    N_BEAMS = 10


class EnvConfig(object):
    MAX_STEPS = 1000
    HISTORY_SIZE = 2
    LIDAR_CONFIG = LidarConfig
    USE_LIDAR_CONFIG = False  # extension: when True, n_beams/spread/distance come from LIDAR_CONFIG


class GameConfig(object):
    BOUNDS = (600, 600)
    SPEED = 10  # multiplier on base_dt = 0.1 (game.py:27,194)
    FPS = 1000  # reference: pygame clock.tick cap (game.py:195); no meaning here, kept for script compatibility
    DEBUG = False


# LiDAR's own constructor defaults (models.py:29) — what the reference actually uses
LIDAR_DEFAULT_N_BEAMS = 10
LIDAR_DEFAULT_SPREAD = 90
LIDAR_DEFAULT_DISTANCE = 100
BASE_DT = 0.1        # ShipGame.base_dt, game.py:27
SPACE_DAMPING = 0.4  # game.py:270
N_GOALS = 5          # game.py:17
