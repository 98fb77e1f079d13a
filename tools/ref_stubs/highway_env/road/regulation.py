from highway_env.road.road import Road


class RegulatedRoad(Road):
    pass
