"""dMacroLane / dMacroForwardLayer on the reference's import path (road.lane.dmacro_lane; reference
dmacro_lane.py:13-309): the drop-in operator, backed by dhts_macro_step_fwd / dhts_macro_step_bwd."""
import numpy as np
import torch as th

from dhts import ops
from road.lane._macro_lane import MacroLane


class dMacroLane(MacroLane):
    """Macroscopic lane differentiated with the analytic per-step Jacobians (the tape the kernel writes)."""

    class dLane:
        """One step's Jacobian tape.  `dqs` gives it in the reference's layout [num_cell][3][2][2]
        (dqs[a, k] = d(next cell a) / d(cell a - 1 + k)); on the device it is [3][Np][4]."""

        def __init__(self, num_cell, tape=None):
            self.num_cell = num_cell
            self.tape = tape

        @property
        def dqs(self):
            n = self.num_cell
            npad = (n + 63) // 64 * 64
            t = self.tape.reshape(3, npad, 4)[:, :n, :]
            return t.permute(1, 0, 2).reshape(n, 3, 2, 2).cpu().numpy()

    def __init__(self, id, lane_length, speed_limit, cell_length):
        super().__init__(id, lane_length, speed_limit, cell_length)
        self.d_lane = []

    def clear(self):
        super().clear()


class dMacroForwardLayer(th.autograd.Function):
    """(lane, r[N+2], y[N+2], delta_time) -> (nr[N], ny[N]); backward (g_nr, g_ny) -> (None, g_r[N+2], g_y[N+2], None).

    As in the reference, u and u_eq of the cells are taken from the lane (they are functions of its (r, y)),
    the lane receives this step's tape (lane.d_lane) and the call is meant to come from lane.forward().
    A CFL violation raises AssertionError (reference _macro_lane.py:141-146)."""

    @staticmethod
    def forward(ctx, lane, r, y, delta_time):
        n = lane.num_cell
        assert r.shape[0] == n + 2 and y.shape[0] == n + 2, "Cell number mismatch"
        desc = ops.macro_desc(1, n, delta_time, lane.cell_length, lane.speed_limit)
        cur = lane._curr.t
        ghost = lane._ghost_tensor()
        ghost[0, 0, 0], ghost[0, 0, 1] = r[0].detach(), y[0].detach()
        ghost[0, 1, 0], ghost[0, 1, 1] = r[-1].detach(), y[-1].detach()
        src_r = getattr(lane, "_left_source_r", None)
        if src_r is not None:            # a source lane's inflow cell: Python floats in the reference, double in the operator
            lo, hi = np.array([src_r], np.float64).view(np.float32)
            ghost[0, 0] = th.tensor([float("nan"), 0.0, 0.0, 0.0], dtype=th.float32).to(ghost.device)
            ghost[0, 0, 2:] = th.from_numpy(np.array([lo, hi], np.float32)).to(ghost.device)
        tape = th.empty(ops.macro_step_tape_numel(desc), dtype=th.float32, device=r.device)
        err = ops.new_error_record(r.device)
        nr, ny, nu, nq = ops.macro_step_fwd(desc, r[1:-1].detach().reshape(1, n), y[1:-1].detach().reshape(1, n),
                                            cur["u"].detach().reshape(1, n), cur["q"].detach().reshape(1, n), ghost,
                                            tape=tape, err=err)
        ops.raise_on_fault(err)
        lane.d_lane.append(dMacroLane.dLane(n, tape))
        ctx.desc, ctx.tape = desc, tape
        nr, ny = nr[0], ny[0]
        lane._step_glue = (nr, ny, nu[0], nq[0])
        return nr, ny

    @staticmethod
    def backward(ctx, grad_nr, grad_ny):
        desc = ctx.desc
        n = desc.n_cells
        err = ops.new_error_record(grad_nr.device)
        g_r, g_y, g_ghost = ops.macro_step_bwd(desc, ctx.tape, grad_nr.contiguous().reshape(1, n),
                                               grad_ny.contiguous().reshape(1, n), err=err)
        ops.raise_on_fault(err)
        gg = g_ghost[0].float()
        grad_r = th.cat([gg[0, 0:1], g_r[0], gg[1, 0:1]])
        grad_y = th.cat([gg[0, 1:2], g_y[0], gg[1, 1:2]])
        return None, grad_r, grad_y, None
