"""MacroLane on the reference's import path (road.lane._macro_lane; reference _macro_lane.py:14-352).

The reference keeps one Python object per cell; here a lane is four float32 device arrays (r, y, u, u_eq) of
length num_cell plus two ghost cells, and `curr_cell[i].state.q.r`-style access is a lazy view over them.
One step is one call of the HIP operator (road.lane.dmacro_lane.dMacroForwardLayer -> dhts_macro_step_fwd).
"""
import math

import torch as th

from dhts import device, ops
from model.macro._arz import ARZ
from road.lane._base_lane import BaseLane


class _StateFromRU(th.autograd.Function):
    """(r, u) -> (y, u_eq): FullQ.set_r_u of every cell in one kernel (dhts_macro_state_from_ru)."""

    @staticmethod
    def forward(ctx, r, u, u_max):
        r, u = r.contiguous(), u.contiguous()
        y, q = ops.macro_state_from_ru(r, u, u_max)
        ctx.save_for_backward(r, u)
        ctx.u_max = u_max
        ctx.mark_non_differentiable(q)
        return y, q

    @staticmethod
    def backward(ctx, g_y, _g_q):
        r, u = ctx.saved_tensors
        g_r = th.zeros_like(r)
        g_u = ops.macro_state_from_ru_bwd(r, u, g_y.contiguous(), g_r, ctx.u_max)
        return g_r, g_u, None


class _SpeedTap(th.autograd.Function):
    """(r, y) -> u of FullQ.set_r_y.  The value comes from the step kernel (it computes u for the next step
    anyway); the adjoint is dhts_macro_u_tap_bwd."""

    @staticmethod
    def forward(ctx, r, y, u_value, u_max):
        ctx.save_for_backward(r.detach().contiguous(), y.detach().contiguous())
        ctx.u_max = u_max
        return u_value.clone()

    @staticmethod
    def backward(ctx, g_u):
        r, y = ctx.saved_tensors
        g_r, g_y = th.zeros_like(r), th.zeros_like(y)
        ops.macro_u_tap_bwd(r, y, g_u.contiguous(), g_r, g_y, ctx.u_max)
        return g_r, g_y, None, None


class _QView:
    def __init__(self, arrs, i):
        self._a, self._i = arrs, i

    r = property(lambda s: s._a.get("r")[s._i], lambda s, v: s._a.put("r", s._i, v))
    y = property(lambda s: s._a.get("y")[s._i], lambda s, v: s._a.put("y", s._i, v))


class _StateView:
    """curr_cell[i].state: q.r, q.y, u, u_eq as 0-dim views of the lane arrays."""

    def __init__(self, arrs, i, u_max):
        self._a, self._i, self.u_max = arrs, i, u_max
        self.q = _QView(arrs, i)

    u = property(lambda s: s._a.get("u")[s._i], lambda s, v: s._a.put("u", s._i, v))
    u_eq = property(lambda s: s._a.get("q")[s._i], lambda s, v: s._a.put("q", s._i, v))

    def set_r_u(self, r, u, u_max):
        y, q = _StateFromRU.apply(device.as_f32(r).reshape(1), device.as_f32(u).reshape(1), float(u_max))
        for k, v in (("r", r), ("y", y[0]), ("u", u), ("q", q[0])):
            self._a.put(k, self._i, v)

    def set_r_y(self, r, y, u_max):
        for k, v in (("r", r), ("y", y), ("u", ARZ.compute_u(device.as_f32(r), device.as_f32(y), u_max)),
                     ("q", ARZ.compute_u_eq(device.as_f32(r), u_max))):
            self._a.put(k, self._i, v)


class _Arrays:
    """The four state arrays of one lane buffer (current or next)."""

    def __init__(self, n, u_max):
        dev = device.get()
        self.t = {"r": th.zeros(n, device=dev), "y": th.zeros(n, device=dev),
                  "u": th.full((n,), float(u_max), device=dev), "q": th.full((n,), float(u_max), device=dev)}

    def get(self, k):
        return self.t[k]

    def put(self, k, i, v):
        # out-of-place so that autograd history of the other cells survives
        self.t[k] = self.t[k].index_put((th.tensor(i, device=self.t[k].device),), device.as_f32(v).reshape(()))

    def set_all(self, r, y, u, q):
        self.t = {"r": r, "y": y, "u": u, "q": q}


class _CellSeq:
    def __init__(self, lane, arrs):
        self._lane, self._a = lane, arrs

    def __len__(self):
        return self._lane.num_cell

    def __getitem__(self, i):
        n = self._lane.num_cell
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        dx = self._lane.cell_length
        cell = MacroLane.Cell(dx * i, dx * (i + 1), self._lane.speed_limit)
        cell.state = _StateView(self._a, i, self._lane.speed_limit)
        return cell

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class MacroLane(BaseLane):

    class Cell:
        """start / end along the lane and the ARZ state (r, y, u, u_eq)."""

        def __init__(self, start, end, speed_limit):
            self.start = start
            self.end = end
            self.state = ARZ.FullQ(speed_limit)

    def __init__(self, id, lane_length, speed_limit, cell_length):
        super().__init__(id, lane_length, speed_limit)
        self.num_cell = math.ceil(self.length / cell_length)
        assert self.num_cell > 0, "Number of cells in a road must be larger than 0."
        self.cell_length = self.length / self.num_cell
        self._curr = _Arrays(self.num_cell, speed_limit)
        self._next = _Arrays(self.num_cell, speed_limit)
        self.curr_cell = _CellSeq(self, self._curr)
        self.next_cell = _CellSeq(self, self._next)
        # ghost cells, used when no lane is connected; by default no flow (r = 0, u = u_max)
        self.leftmost_cell = MacroLane.Cell(0, 0, speed_limit)
        self.rightmost_cell = MacroLane.Cell(self.cell_length, self.cell_length, speed_limit)
        self.riemann_solution = None
        self.flux_capacitor = {}
        self.bdry_callback = None
        self.bdry_callback_args = {"lane": self}
        self.d_lane = []

    def is_macro(self):
        return True

    def is_micro(self):
        return False

    def which(self, pos):
        return math.floor(pos / self.cell_length)

    # ---- ghosts --------------------------------------------------------------------------------------------
    def _ghost(self, r, u):
        r1, u1 = device.as_f32(r).reshape(1), device.as_f32(u).reshape(1)
        y, q = _StateFromRU.apply(r1, u1, float(self.speed_limit))
        fq = ARZ.FullQ(self.speed_limit)
        fq.q = ARZ.Q(r1[0], y[0])
        fq.u, fq.u_eq = u1[0], q[0]
        return fq

    def set_leftmost_cell(self, r, u):
        self.leftmost_cell.state = self._ghost(r, u)
        # A boundary cell of plain Python floats stays one in the reference (dMacroLane.decell leaves floats alone) and its Riemann
        # solve reads it in double; the one such cell of the itscp networks is a source lane's inflow (r, u_eq(r)), which the step
        # operator takes in double (include/dhts.h, dhts_macro_step_fwd): remembered here for dMacroForwardLayer
        self._left_source_r = float(r) if (isinstance(r, float) and isinstance(u, float)
                                           and u == ARZ.compute_u_eq(r, self.speed_limit)) else None

    def set_rightmost_cell(self, r, u):
        self.rightmost_cell.state = self._ghost(r, u)

    def get_leftmost_cell(self):
        return self.leftmost_cell

    def get_rightmost_cell(self):
        return self.rightmost_cell

    def get_left_cell(self, id):
        return self.get_leftmost_cell() if id == 0 else self.curr_cell[id - 1]

    def get_right_cell(self, id):
        return self.get_rightmost_cell() if id == self.num_cell - 1 else self.curr_cell[id + 1]

    def _ghost_tensor(self):
        """[1][2][4] float32: (left, right) x (r, y, u, u_eq), detached values for the kernel."""
        rows = []
        for c in (self.leftmost_cell, self.rightmost_cell):
            s = c.state
            rows.append(th.stack([device.as_f32(v).reshape(()) for v in (s.q.r, s.q.y, s.u, s.u_eq)]))
        return th.stack(rows).detach().reshape(1, 2, 4).contiguous()

    # ---- state vectors ----------------------------------------------------------------------------------------
    def _check_len(self, *vs):
        for v in vs:
            assert len(v) == self.num_cell, "Cell number mismatch"

    def set_state_vector_u(self, rv, uv):
        self._check_len(rv, uv)
        r, u = device.as_f32(rv), device.as_f32(uv)
        y, q = _StateFromRU.apply(r, u, float(self.speed_limit))
        self._curr.set_all(r, y, u, q)

    def set_state_vector_y(self, rv, yv):
        self._check_len(rv, yv)
        r, y = device.as_f32(rv), device.as_f32(yv)
        self._curr.set_all(r, y, ARZ.compute_u(r, y, self.speed_limit), ARZ.compute_u_eq(r, self.speed_limit))

    def set_next_state_vector_u(self, rv, uv):
        self._check_len(rv, uv)
        r, u = device.as_f32(rv), device.as_f32(uv)
        y, q = _StateFromRU.apply(r, u, float(self.speed_limit))
        self._next.set_all(r, y, u, q)

    def set_next_state_vector_y(self, rv, yv):
        self._check_len(rv, yv)
        r, y = device.as_f32(rv), device.as_f32(yv)
        glue = getattr(self, "_step_glue", None)
        if (glue is not None and isinstance(rv, th.Tensor) and isinstance(yv, th.Tensor)
                and glue[0].data_ptr() == rv.data_ptr() and glue[1].data_ptr() == yv.data_ptr()):
            # (r, y) are the step kernel's outputs: it already evaluated u, u_eq for them
            u = _SpeedTap.apply(r, y, glue[2], float(self.speed_limit))
            q = glue[3]
        else:
            u, q = ARZ.compute_u(r, y, self.speed_limit), ARZ.compute_u_eq(r, self.speed_limit)
        self._step_glue = None
        self._next.set_all(r, y, u, q)

    def get_state_vector(self):
        t = self._curr.t
        return t["r"], t["y"], t["u"]

    def get_next_state_vector(self):
        t = self._next.t
        return t["r"], t["y"], t["u"]

    def update_state(self):
        self._curr.set_all(*(self._next.t[k] for k in ("r", "y", "u", "q")))

    def add_flux_capacitor(self, next_lane_id, increment):
        self.flux_capacitor[next_lane_id] = self.flux_capacitor.get(next_lane_id, 0.0) + increment

    def clear(self):
        for a in (self._curr, self._next):
            fresh = _Arrays(self.num_cell, self.speed_limit)
            a.t = fresh.t
        self.flux_capacitor.clear()

    # ---- one step ---------------------------------------------------------------------------------------------
    def vectorize_input(self):
        """(r, y) of the lane padded with the ghosts: the operator's differentiable inputs [N + 2]."""
        t = self._curr.t
        l, rr = self.leftmost_cell.state, self.rightmost_cell.state
        as1 = lambda v: device.as_f32(v).reshape(1)     # noqa: E731
        cr = th.cat([as1(l.q.r), t["r"], as1(rr.q.r)])
        cy = th.cat([as1(l.q.y), t["y"], as1(rr.q.y)])
        return cr, cy

    def forward(self, delta_time):
        from road.lane.dmacro_lane import dMacroForwardLayer
        cr, cy = self.vectorize_input()
        nr, ny = dMacroForwardLayer.apply(self, cr, cy, delta_time)
        self.set_next_state_vector_y(nr, ny)
