"""Minimal ``Box`` space (gym is not a dependency of the engine).

Mirrors what the hot path uses of ``gym.spaces.Box`` (gym 0.21, pinned at
/root/reference/pyproject.toml:18): float32-typed ``low`` / ``high`` arrays (that rounding is part
of the reference's numerics: quadrotor.py:731-733,791; constraints.py:363-364), ``shape``,
``sample()``, ``seed()``, ``contains()``.
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            low = np.asarray(low)
            shape = low.shape
        shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), shape).astype(self.dtype)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), shape).astype(self.dtype)
        self.shape = shape
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self, n=None):
        """One sample (shape) or a batch (n, *shape).  Unbounded dims draw from a normal."""
        shape = self.shape if n is None else (n,) + self.shape
        finite = np.isfinite(self.low) & np.isfinite(self.high) & (np.abs(self.low) < 1e30) & (np.abs(self.high) < 1e30)
        lo = np.where(finite, self.low, 0.0)
        hi = np.where(finite, self.high, 1.0)
        u = self._rng.uniform(size=shape) * (hi - lo) + lo
        g = self._rng.normal(size=shape)
        return np.where(finite, u, g).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return 'Box({}, {}, {}, {})'.format(self.low.min(), self.high.max(), self.shape, self.dtype)
