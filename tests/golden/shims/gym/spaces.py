import numpy as np


class Discrete(object):
    def __init__(self, n):
        self.n = n

    def contains(self, x):
        if isinstance(x, (int, np.integer)):
            return 0 <= int(x) < self.n
        if isinstance(x, np.ndarray) and x.dtype.kind in "iu" and x.shape == ():
            return 0 <= int(x) < self.n
        return False


class Box(object):
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.low = (low + np.zeros(shape)).astype(dtype)
        self.high = (high + np.zeros(shape)).astype(dtype)
