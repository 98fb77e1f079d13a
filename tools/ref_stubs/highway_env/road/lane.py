import numpy as np


class LineType:
    NONE, STRIPED, CONTINUOUS, CONTINUOUS_LINE = 0, 1, 2, 3


class AbstractLane:
    DEFAULT_WIDTH = 4


class StraightLane(AbstractLane):
    """Straight segment: length = |end - start|; position(s, lateral) along / across it."""

    def __init__(self, start, end, width=AbstractLane.DEFAULT_WIDTH, line_types=None, forbidden=False, speed_limit=20, priority=0):
        self.start = np.array(start, dtype=float)
        self.end = np.array(end, dtype=float)
        self.width = width
        self.length = float(np.linalg.norm(self.end - self.start))
        self.heading = float(np.arctan2(self.end[1] - self.start[1], self.end[0] - self.start[0]))
        self.direction = (self.end - self.start) / self.length
        self.direction_lateral = np.array([-self.direction[1], self.direction[0]])
        self.line_types = line_types
        self.speed_limit = speed_limit

    def position(self, longitudinal, lateral):
        return self.start + longitudinal * self.direction + lateral * self.direction_lateral

    def heading_at(self, longitudinal):
        return self.heading
