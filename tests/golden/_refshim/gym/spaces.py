import numpy as np


class Box:
  def __init__(self, low, high, shape=None, dtype=np.float32):
    self.low, self.high, self.shape, self.dtype = low, high, shape, dtype
