import numpy as np


class Box:
    def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
        if shape is None:
            low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
            shape = low.shape
        else:
            low, high = np.full(shape, low, dtype=dtype), np.full(shape, high, dtype=dtype)
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)
