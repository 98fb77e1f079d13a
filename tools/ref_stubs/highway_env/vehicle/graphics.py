class VehicleGraphics:
    pass
