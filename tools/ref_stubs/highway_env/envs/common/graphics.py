class EnvViewer:
    def __init__(self, env, config=None):
        self.env = env


class ObservationGraphics:
    pass


class EventHandler:
    pass
