class BaseLane:
    """Common lane record: id, length, speed limit and the prev / next connectivity dictionaries
    (reference road/lane/_base_lane.py:7-53)."""

    def __init__(self, id, length, speed_limit):
        self.id = id
        self.length = length
        self.speed_limit = speed_limit
        self.next_lane = {}
        self.prev_lane = {}

    def is_macro(self):
        raise NotImplementedError()

    def is_micro(self):
        raise NotImplementedError()

    def forward(self, delta_time):
        raise NotImplementedError()

    def update_state(self):
        raise NotImplementedError()

    def clear(self):
        raise NotImplementedError()

    def add_prev_lane(self, lane):
        self.prev_lane[lane.id] = lane

    def add_next_lane(self, lane):
        self.next_lane[lane.id] = lane

    def num_prev_lane(self):
        return len(self.prev_lane)

    def num_next_lane(self):
        return len(self.next_lane)

    def has_prev_lane(self):
        return self.num_prev_lane() > 0

    def has_next_lane(self):
        return self.num_next_lane() > 0
