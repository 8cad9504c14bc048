"""Minimal observation/action space descriptors (gym is not a dependency; same attribute names as gym.spaces)."""
import numpy as np


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n


class Box:
    def __init__(self, low, high, shape, dtype="uint8"):
        self.low, self.high = low, high
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and x.dtype == self.dtype

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, self.dtype)

    def __eq__(self, other):
        return isinstance(other, Box) and (self.low, self.high, self.shape, self.dtype) == (other.low, other.high, other.shape, other.dtype)
