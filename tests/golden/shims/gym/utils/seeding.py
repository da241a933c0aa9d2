import numpy as np


def np_random(seed=None):
    """gym.utils.seeding.np_random: (a private RandomState, the seed it was given)."""
    rng = np.random.RandomState()
    rng.seed(seed)
    return rng, seed
