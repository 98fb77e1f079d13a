"""Soft switch used by boundary / signal / reward glue.  Mirrors dmath/operation.py:3-30 of the reference
(sigmoid(clamp(value * constant, min, max))); plain torch, runs on whatever device `value` lives on."""
import torch as th


def sigmoid(value, constant, min=-16.0, max=16.0):
    if not isinstance(value, th.Tensor):
        value = th.tensor(value)
    scaled = value * (constant if isinstance(constant, th.Tensor) else float(constant))
    return th.sigmoid(th.clamp(scaled, min, max))
