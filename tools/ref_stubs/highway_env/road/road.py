class RoadNetwork:
    def __init__(self):
        self.graph = {}

    def add_lane(self, _from, _to, lane):
        self.graph.setdefault(_from, {}).setdefault(_to, []).append(lane)


class Road:
    def __init__(self, network=None, vehicles=None, road_objects=None, np_random=None, record_history=False):
        self.network = network
        self.vehicles = vehicles or []
        self.np_random = np_random
