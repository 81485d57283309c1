"""The sliver of gym.spaces the reference's envs expose (gym itself is not a dependency)."""
import numpy as np


class Box:
  def __init__(self, low, high, shape, dtype=np.float32):
    self.shape = tuple(shape)
    self.dtype = np.dtype(dtype)
    self.low = np.full(self.shape, low, dtype=self.dtype)
    self.high = np.full(self.shape, high, dtype=self.dtype)
    self._rng = np.random.default_rng()

  def seed(self, seed=None):
    self._rng = np.random.default_rng(seed)

  def sample(self):
    lo = np.where(np.isfinite(self.low), self.low, -1.0)
    hi = np.where(np.isfinite(self.high), self.high, 1.0)
    return self._rng.uniform(lo, hi).astype(self.dtype)

  def contains(self, x):
    x = np.asarray(x)
    return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

  def __repr__(self):
    return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'
