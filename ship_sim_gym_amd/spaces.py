"""gym.spaces stand-ins used only when gym is not importable (it is absent from this image).

The reference declares ``action_space = Discrete(3)`` and ``observation_space = Box(low=0, high=max(bounds),
shape=(n_states*H,), dtype=np.uint8)`` (ship_env.py:19,48).  When gym/gymnasium is installed the real classes
are used so SB / RLlib type checks pass; otherwise these duck-typed equivalents are.
"""
import numpy as np

try:  # pragma: no cover - depends on the environment
    from gym.spaces import Box, Discrete  # type: ignore
    HAVE_GYM = True
except Exception:  # gym missing (or broken)
    try:
        from gymnasium.spaces import Box, Discrete  # type: ignore
        HAVE_GYM = True
    except Exception:
        HAVE_GYM = False

        class Discrete(object):
            def __init__(self, n):
                self.n = int(n)
                self.shape = ()
                self.dtype = np.dtype(np.int64)
                self._rng = np.random.RandomState()

            def seed(self, seed=None):
                self._rng = np.random.RandomState(seed)
                return [seed]

            def sample(self):
                return int(self._rng.randint(self.n))

            def contains(self, x):
                if isinstance(x, (int, np.integer)):
                    v = int(x)
                elif isinstance(x, np.ndarray) and x.dtype.kind in "iu" and x.shape == ():
                    v = int(x)
                else:
                    return False
                return 0 <= v < self.n

            def __repr__(self):
                return "Discrete(%d)" % self.n

            def __eq__(self, other):
                return isinstance(other, Discrete) and other.n == self.n

        class Box(object):
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.dtype = np.dtype(dtype)
                self.shape = tuple(shape) if shape is not None else np.shape(low)
                # gym 0.10.9 casts the bounds to the declared dtype (600 -> uint8 88), SURVEY.md §8b
                with np.errstate(over="ignore"):
                    self.low = np.full(self.shape, low).astype(self.dtype)
                    self.high = np.full(self.shape, high).astype(self.dtype)
                self._rng = np.random.RandomState()

            def seed(self, seed=None):
                self._rng = np.random.RandomState(seed)
                return [seed]

            def sample(self):
                return self._rng.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

            def __repr__(self):
                return "Box%s" % (self.shape,)

            def __eq__(self, other):
                return (isinstance(other, Box) and self.shape == other.shape and np.allclose(self.low, other.low)
                        and np.allclose(self.high, other.high))
