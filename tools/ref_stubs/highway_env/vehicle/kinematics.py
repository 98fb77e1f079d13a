class Vehicle:
    def __init__(self, road, position, heading=0, speed=0):
        self.road, self.position, self.heading, self.speed = road, position, heading, speed
