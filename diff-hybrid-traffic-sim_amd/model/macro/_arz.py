"""ARZ helpers on the reference's import path (model.macro._arz).

The per-interface arithmetic of the hot path (Riemann solve, Jacobians) lives in the HIP kernels
(csrc/arz_device.hpp); this module keeps the reference's *names* so host code written against
model/macro/_arz.py keeps working:
  - compute_y / compute_u / compute_u_eq / compute_u_eq_prime / compute_r_from_u_eq  (_arz.py:121-149):
    closed forms, evaluated with torch ops on tensors (any device) or Python floats;
  - Q / FullQ (_arz.py:10-115): light records used for ghost cells and for the scalar entry points;
  - riemann_solve (_arz.py:212-332): one interface through dhts_arz_interface_batch on the GPU.
"""
import math

import torch as th

GAMMA = 0.5
EPSILON = 1e-5


def _is_t(*xs):
    return any(isinstance(x, th.Tensor) for x in xs)


class ARZ:

    class Q:
        def __init__(self, r=0, y=0):
            self.r = r
            self.y = y

        @staticmethod
        def from_r_y(r, y):
            return ARZ.Q(r, y)

        @staticmethod
        def from_r_u(r, u, u_max):
            return ARZ.Q(r, ARZ.compute_y(r, u, u_max))

        def __mul__(self, c):
            return ARZ.Q(self.r * c, self.y * c)

        def __add__(self, o):
            return ARZ.Q(self.r + o.r, self.y + o.y)

        def __sub__(self, o):
            return ARZ.Q(self.r - o.r, self.y - o.y)

        def clear(self):
            self.r = 0
            self.y = 0

    class FullQ:
        def __init__(self, u_max):
            self.q = ARZ.Q()
            self.u_max = u_max
            self.u = u_max
            self.u_eq = u_max

        @staticmethod
        def from_q(q, u_max):
            fq = ARZ.FullQ(u_max)
            fq.set_r_y(q.r, q.y, u_max)
            return fq

        @staticmethod
        def from_r_u(r, u, u_max):
            fq = ARZ.FullQ(u_max)
            fq.set_r_u(r, u, u_max)
            return fq

        def set_r_u(self, r, u, u_max):
            self.u_max = u_max
            self.u = u
            self.q = ARZ.Q.from_r_u(r, u, u_max)
            self.u_eq = ARZ.compute_u_eq(r, u_max)

        def set_r_y(self, r, y, u_max):
            self.u_max = u_max
            self.u = ARZ.compute_u(r, y, u_max)
            self.q = ARZ.Q(r, y)
            self.u_eq = ARZ.compute_u_eq(r, u_max)

        def flux_r(self):
            return self.q.r * self.u

        def flux_y(self):
            return self.q.y * self.u

        def flux(self):
            return ARZ.Q(self.flux_r(), self.flux_y())

        def lambda_0(self):
            return self.u + self.q.r * ARZ.compute_u_eq_prime(self.q.r, self.u_max)

        def lambda_1(self):
            return self.u

        def clear(self):
            self.q.clear()
            self.u = self.u_max
            self.u_eq = self.u_max

    # ---- closed forms -----------------------------------------------------------------------------------
    @staticmethod
    def compute_u_eq(r, u_max, gamma=GAMMA):
        if _is_t(r):
            return u_max * (1.0 - th.pow(th.clamp(r, min=0.0) + EPSILON, gamma))
        return u_max * (1.0 - math.pow(max(r, 0.0) + EPSILON, gamma))

    @staticmethod
    def compute_u_eq_prime(r, u_max, gamma=GAMMA):
        if _is_t(r):
            return -u_max * gamma * th.pow(th.clamp(r, min=EPSILON), gamma - 1)
        return -u_max * gamma * math.pow(max(r, EPSILON), gamma - 1)

    @staticmethod
    def compute_y(r, u, u_max):
        return r * (u - ARZ.compute_u_eq(r, u_max))

    @staticmethod
    def compute_u(r, y, u_max):
        if _is_t(r, y):
            r = th.as_tensor(r)
            rc = th.where(r < EPSILON, th.full_like(r, EPSILON), r)       # hard gate like the reference's max()
            return y / rc + ARZ.compute_u_eq(rc, u_max)
        rc = max(r, EPSILON)
        return y / rc + ARZ.compute_u_eq(rc, u_max)

    @staticmethod
    def compute_r_from_u_eq(u_eq, u_max, gamma=GAMMA):
        u_max = max(u_max, EPSILON)
        gamma = max(gamma, EPSILON)
        return (1.0 - u_eq / u_max) ** (1.0 / gamma)

    # ---- one interface on the device ----------------------------------------------------------------------
    class Riemann:
        def __init__(self):
            self.speed0 = 0
            self.speed1 = 0
            self.Q_0 = ARZ.FullQ(0)
            self.case_ind = -1      # 0 = Q_L, 1 = Q_M, 2 = Q_C
            self._detail = None

    @staticmethod
    def _solve_pair(Q_L, Q_R, u_max):
        from dhts import device, ops
        row = [float(Q_L.q.r), float(Q_L.q.y), float(Q_L.u), float(Q_L.u_eq),
               float(Q_R.q.r), float(Q_R.q.y), float(Q_R.u), float(Q_R.u_eq), float(u_max)]
        return ops.arz_interface_batch(th.tensor([row], dtype=th.float64, device=device.get()))

    @staticmethod
    def riemann_solve(Q_L, Q_R, u_max):
        """State at the interface between a left and a right cell: case index, Q_0 and the two wave speeds
        (reference _arz.py:212-332)."""
        out = ARZ._solve_pair(Q_L, Q_R, u_max)
        rs = ARZ.Riemann()
        rs.case_ind = int(out["case"][0])
        rs.speed0, rs.speed1 = (float(x) for x in out["speed"][0].tolist())
        q0 = out["q0"][0].tolist()
        rs.Q_0 = ARZ.FullQ(u_max)
        rs.Q_0.q = ARZ.Q(q0[0], q0[1])
        rs.Q_0.u, rs.Q_0.u_eq = q0[2], q0[3]
        rs._detail = out
        return rs
