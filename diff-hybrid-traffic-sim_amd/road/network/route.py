class MacroRoute:
    """One-to-one next / previous assignment between connected macro lanes for one step
    (reference road/network/route.py:3-37)."""

    def __init__(self):
        self.next_lane_dict = {}
        self.prev_lane_dict = {}

    def get_next_lane(self, lane_id):
        return self.next_lane_dict.get(lane_id, -1)

    def get_prev_lane(self, lane_id):
        return self.prev_lane_dict.get(lane_id, -1)


class MicroRoute:
    """Lane sequence of one vehicle plus a cursor (reference route.py:39-82)."""

    def __init__(self, route, curr_idx=0):
        self.route = route
        self.curr_idx = curr_idx

    def increment_curr_idx(self):
        self.curr_idx += 1

    def route_length(self):
        return len(self.route)

    def curr_lane_id(self):
        return self.route[self.curr_idx]

    def prev_lane_id(self):
        return self.route[self.curr_idx - 1] if self.curr_idx > 0 else -1

    def next_lane_id(self):
        return self.route[self.curr_idx + 1] if self.curr_idx < self.route_length() - 1 else -1
