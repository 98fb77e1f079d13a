import numpy as np

from road.vehicle.vehicle import DEFAULT_VEHICLE_LENGTH, Vehicle


class MicroVehicle(Vehicle):
    """IDM vehicle record (reference road/vehicle/micro_vehicle.py:5-28).  The six IDM parameters
    (accel_max, accel_pref, target_speed, min_space, time_pref, length) are what the kernel keeps in registers."""

    PARAM_NAMES = ("accel_max", "accel_pref", "target_speed", "min_space", "time_pref", "length")

    def __init__(self, id, position, speed, accel_max, accel_pref, target_speed, min_space, time_pref, length, a):
        super().__init__(id, position, speed, length, a)
        self.accel_max = accel_max
        self.accel_pref = accel_pref
        self.target_speed = target_speed
        self.min_space = min_space
        self.time_pref = time_pref

    def params(self):
        return [float(getattr(self, k)) for k in MicroVehicle.PARAM_NAMES]

    @staticmethod
    def default_micro_vehicle(speed_limit):
        """a_max = 1.0 sl, a_pref = 0.8 sl, v_target = 0.9 sl, s0 = 0.1 len, T = 0.1 (reference :31-72)."""
        ln = DEFAULT_VEHICLE_LENGTH
        return MicroVehicle(-1, 0, 0, speed_limit * 1., speed_limit * 0.8, speed_limit * 0.9, ln * 0.1, 0.1, ln, ln)

    @staticmethod
    def random_micro_vehicle(speed_limit):
        """Randomised parameters (reference :75-121): a_max in [1.5, 2] sl, a_pref in [1, 1.5] sl, target speed in
        [0.8, 1.2] sl, min_space in [0.2, 0.4] len, time_pref in [0.2, 0.6]; five np.random.rand() draws in that order."""
        ln = DEFAULT_VEHICLE_LENGTH
        u = [np.random.rand() for _ in range(5)]
        lerp = lambda t, a, b: a + (b - a) * t      # noqa: E731
        return MicroVehicle(-1, 0, 0,
                            lerp(u[0], speed_limit * 1.5, speed_limit * 2.0), lerp(u[1], speed_limit * 1.0, speed_limit * 1.5),
                            lerp(u[2], speed_limit * 0.8, speed_limit * 1.2), lerp(u[3], ln * 0.2, ln * 0.4),
                            lerp(u[4], 0.2, 0.6), ln, ln)
