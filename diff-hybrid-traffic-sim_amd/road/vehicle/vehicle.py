DEFAULT_VEHICLE_LENGTH = 5.0


class Vehicle:
    """Host-side record of a vehicle: id, position, speed, length and the ancillary density carrier `a`
    (reference road/vehicle/vehicle.py:3-17).  On the device a vehicle is a slot of the lane's SoA arrays."""

    def __init__(self, id, position, speed, length, a):
        self.id = id
        self.position = position
        self.speed = speed
        self.length = length
        self.a = a
