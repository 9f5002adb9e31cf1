"""Observation / action spaces.  gymnasium's classes are used when gymnasium is installed;
otherwise two small value-compatible stand-ins (same attributes RLlib-style callers read:
``n``, ``shape``, ``dtype``, ``low``, ``high``, ``sample()``, ``contains()``)."""

from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Discrete  # type: ignore
except Exception:  # gymnasium absent (this image): minimal equivalents

    class Discrete:  # type: ignore[no-redef]
        def __init__(self, n: int, seed: int | None = None):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.int64
            self._rng = np.random.default_rng(seed)

        def sample(self) -> int:
            return int(self._rng.integers(0, self.n))

        def contains(self, v) -> bool:
            try:
                return 0 <= int(v) < self.n and int(v) == v
            except Exception:
                return False

        __contains__ = contains

        def __eq__(self, other) -> bool:
            return isinstance(other, Discrete) and other.n == self.n

        def __repr__(self) -> str:
            return f"Discrete({self.n})"

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape, dtype=np.float32, seed: int | None = None):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, self.dtype)
            self.high = np.full(self.shape, high, self.dtype)
            self._rng = np.random.default_rng(seed)

        def sample(self) -> np.ndarray:
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, v) -> bool:
            v = np.asarray(v)
            return v.shape == self.shape and bool(np.all(v >= self.low) and np.all(v <= self.high))

        __contains__ = contains

        def __eq__(self, other) -> bool:
            return (isinstance(other, Box) and other.shape == self.shape and
                    np.array_equal(other.low, self.low) and np.array_equal(other.high, self.high))

        def __repr__(self) -> str:
            return f"Box({self.low.flat[0]}, {self.high.flat[0]}, {self.shape}, {self.dtype})"
