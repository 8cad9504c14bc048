from collections import OrderedDict

import numpy as np


class Space(object):
    shape = None
    dtype = None


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def contains(self, x):
        return 0 <= int(x) < self.n

    def sample(self):
        return int(np.random.randint(self.n))

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            low, high = np.asarray(low), np.asarray(high)
            shape = low.shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, self.dtype), self.shape).copy()

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = OrderedDict(spaces)
