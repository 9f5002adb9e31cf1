"""Minimal gymnasium stand-in (see ../README.md)."""
import numpy as np

from . import spaces  # noqa: F401
from .spaces import Space  # noqa: F401


class Env:
    metadata: dict = {}
    np_random = None

    def reset(self, *, seed=None, options=None):
        # gymnasium.utils.seeding.np_random: Generator(PCG64(SeedSequence(seed)))
        if seed is not None or self.np_random is None:
            self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def close(self):
        pass
