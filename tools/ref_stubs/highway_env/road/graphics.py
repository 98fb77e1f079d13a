class WorldSurface:
    pass


class RoadGraphics:
    pass
