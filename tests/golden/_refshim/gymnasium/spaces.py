import numpy as np


class Space:
    pass


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self._rng = np.random.default_rng(0)

    def sample(self):
        return int(self._rng.integers(0, self.n))

    def contains(self, v):
        return 0 <= int(v) < self.n


class Box(Space):
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
