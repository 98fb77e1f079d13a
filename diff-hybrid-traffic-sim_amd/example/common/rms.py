import numpy as np


class RunningMean:
    """Mean over the last `size` samples (reference example/common/rms.py:3-23).  `prefix_means` returns, for a block
    of new samples appended one at a time, the running mean after each append -- what the reference obtains by calling
    update() + mean() once per cell / vehicle."""

    def __init__(self, size):
        self.size = size
        self.data = np.array([], dtype=np.float32)

    def update(self, data):
        data = np.asarray(data, dtype=np.float32)
        if data.ndim == 0:
            data = data.reshape(1)
        self.data = np.concatenate([self.data, data])[-self.size:]

    def mean(self):
        return np.mean(self.data)

    def std(self):
        return np.clip(np.std(self.data), 1e-4, None)

    def prefix_means(self, block):
        block = np.asarray(block, dtype=np.float32).reshape(-1)
        n_old, n_new = len(self.data), len(block)
        allv = np.concatenate([self.data, block]).astype(np.float64)
        csum = np.concatenate([[0.0], np.cumsum(allv)])
        ends = np.arange(n_old + 1, n_old + n_new + 1)
        starts = np.maximum(ends - self.size, 0)
        means = (csum[ends] - csum[starts]) / (ends - starts)
        self.data = allv[-self.size:].astype(np.float32)
        return means.astype(np.float32)
