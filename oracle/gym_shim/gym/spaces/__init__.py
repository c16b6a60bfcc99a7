"""gym.spaces.{Box, Discrete}: only the attributes the reference scripts read."""
import numpy as np

from gym.utils import seeding


class Space:
    def __init__(self, shape=None, dtype=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._np_random, seed = seeding.np_random(seed)
        return [seed]


class Discrete(Space):
    def __init__(self, n):
        assert n >= 0
        self.n = n
        super().__init__((), np.int64)

    def sample(self):
        return self.np_random.randint(self.n)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is not None:
            low = np.broadcast_to(low, shape).copy()
            high = np.broadcast_to(high, shape).copy()
        self.low, self.high = low, high
        super().__init__(low.shape, dtype)

    def sample(self):
        return self.np_random.uniform(low=self.low, high=self.high, size=self.shape).astype(self.dtype)
