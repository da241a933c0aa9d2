"""gym 0.10.9 stand-in, test infrastructure only (see ../README.md)."""
from . import spaces, utils  # noqa: F401


class Env(object):
    metadata = {'render.modes': []}
    reward_range = (-float('inf'), float('inf'))
    action_space = None
    observation_space = None
