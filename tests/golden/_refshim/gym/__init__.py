"""Minimal stand-in for gym 0.23 (golden-vector generation only; see ../README.md)."""
from . import spaces  # noqa: F401


class Env:
  pass


class Wrapper(Env):
  def __init__(self, env):
    self.env = env

  def __getattr__(self, name):
    if name.startswith('_'):
      raise AttributeError(name)
    return getattr(self.env, name)
