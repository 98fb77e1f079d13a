"""dhts -- MI355X-native differentiable traffic stepper (ARZ cell stencil + IDM ODE, forward + adjoint).

Host-side Python over libdhts.so (hand-written HIP for gfx950, C ABI in include/dhts.h).
"""
from . import _lib, dist, ops  # noqa: F401
from ._lib import DhtsError  # noqa: F401
from .ops import (MacroRollout, MicroRollout, macro_rollout, micro_rollout)  # noqa: F401
