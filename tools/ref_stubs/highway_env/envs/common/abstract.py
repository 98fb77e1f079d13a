import numpy as np


class AbstractEnv:
    """Only what ItscpEnv touches: a config dict, np_random, viewer slot, update_metadata()."""

    def __init__(self, config=None):
        self.config = self.default_config()
        if config:
            self.config.update(config)
        self.np_random = np.random.RandomState(0)
        self.viewer = None
        self.road = None
        self.time = self.steps = 0
        self.done = False

    @classmethod
    def default_config(cls):
        return {}

    def update_metadata(self):
        pass


Action = object
