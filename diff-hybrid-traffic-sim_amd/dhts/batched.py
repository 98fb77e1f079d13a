"""Macro road networks of ANY size, one operator call per step for all lanes (round 4).

The fused network kernels (dhts_net_macro_rollout_*) keep a replica in ONE workgroup: cells + lanes <= 1024.  The reference
builds grids of any size (example/control/itscp/_env.py:221-439: `--n_intersection=3 --n_lane=3` has 360 lanes and ~2 100
cells); until this round such a network ran lane by lane -- one dhts_macro_step_fwd launch plus host glue per lane and step,
~50 s per differentiable episode.  This module is the tier between the two: the network's lanes are the BATCH of the
straight-lane step operator (lanes grouped by (cells, cell length); dhts_macro_step_fwd / _bwd, the same HIP kernels and tape as
dMacroForwardLayer), the ghost exchange of a step (road_network.py:79-111, _simulator.py:42-60) is a handful of gathers and
blends over all lanes at once in float32 like the reference's glue, and the queue loss with its RunningMean(100 000)
(_env.py:586-618, rms.py) is evaluated for all steps at once from the state history (a prefix sum over the sample stream).
Device tensors and autograd only; no per-lane Python, no host round trip inside the episode (faults are read once at the end).
The tables are dhts.network.MacroNetworkTables -- the same the fused kernels take.
"""
import numpy as np
import torch

from . import ops

WINDOW = 100_000      # RunningMean(100_000), reference _env.py:122


class _StateFromRU(torch.autograd.Function):
    """(r, u) -> (y, u_eq) of FullQ.set_r_u (dhts_macro_state_from_ru), any 1-D batch."""

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
        g_r = torch.zeros_like(r)
        g_u = ops.macro_state_from_ru_bwd(r, u, g_y.contiguous(), g_r, ctx.u_max)
        return g_r, g_u, None


class _SpeedTap(torch.autograd.Function):
    """(r, y) -> u of FullQ.set_r_y: the value is the step kernel's, the adjoint dhts_macro_u_tap_bwd."""

    @staticmethod
    def forward(ctx, r, y, u_value, u_max):
        ctx.save_for_backward(r.detach().contiguous(), y.detach().contiguous())
        ctx.u_max = u_max
        return u_value.clone()

    @staticmethod
    def backward(ctx, g_u):
        r, y = ctx.saved_tensors
        g_r, g_y = torch.zeros_like(r), torch.zeros_like(y)
        ops.macro_u_tap_bwd(r, y, g_u.contiguous(), g_r, g_y, ctx.u_max)
        return g_r, g_y, None, None


class _LaneBatchStep(torch.autograd.Function):
    """One ARZ step of a batch of lanes with given ghosts: dMacroForwardLayer (reference dmacro_lane.py:13-309) for [B][N] lanes.
    (r, y [B][N]; ghost_ry [B][2][2] = (left, right) x (r, y), differentiable; u, q [B][N] and ghost_uq [B][2][2] = (u, u_eq):
    functions of the former, taken as values) -> nr, ny (differentiable), nu, nq (the kernel's glue for the next state)."""

    @staticmethod
    def forward(ctx, r, y, ghost_ry, u, q, ghost_uq, desc, err):
        ghost = torch.cat([ghost_ry, ghost_uq], dim=-1).contiguous()            # [B][2][4] = (r, y, u, u_eq)
        tape = torch.empty(ops.macro_step_tape_numel(desc), dtype=torch.float32, device=r.device)
        nr, ny, nu, nq = ops.macro_step_fwd(desc, r.contiguous(), y.contiguous(), u.contiguous(), q.contiguous(), ghost, tape=tape, err=err)
        ctx.desc, ctx.tape, ctx.err = desc, tape, err
        ctx.mark_non_differentiable(nu, nq)
        return nr, ny, nu, nq

    @staticmethod
    def backward(ctx, g_nr, g_ny, _g_nu, _g_nq):
        g_r, g_y, g_ghost = ops.macro_step_bwd(ctx.desc, ctx.tape, g_nr.contiguous(), g_ny.contiguous(), err=ctx.err)
        return g_r, g_y, g_ghost.float(), None, None, None, None, None


def _soft(x, c):
    """dmath/operation.py:3-30: sigmoid(clamp(x * c, -16, 16))."""
    return torch.sigmoid(torch.clamp(x * c, -16.0, 16.0))


class BatchedMacroNetwork:
    """MacroNetworkTables on the device, lanes grouped for the batched step operator.  `rollout` = ItscpEnv.step of a `macro`
    network of any size: (reward, queue [T][L])."""

    def __init__(self, tables, device):
        t = tables
        self.t, self.device = t, device
        L, C = t.n_lanes, t.n_cells
        up = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)      # noqa: E731
        off, n = np.asarray(t.lane_off, dtype=np.int64), np.asarray(t.lane_ncell, dtype=np.int64)
        self.first, self.last = up(off, torch.long), up(off + n - 1, torch.long)
        self.left_src, self.left_gate, self.right_src = (up(x, torch.long) for x in (t.left_src, t.left_gate, t.right_src))      # [T][L]
        self.schedule = up(t.schedule, torch.float64)                                                                                # [T][L]
        self.kind, self.inter = up(t.sig_kind, torch.long), up(t.inter, torch.long)
        # groups of lanes with the same number of cells and the same cell length: one operator call each per step
        keys = {}
        for l in range(L):
            keys.setdefault((int(n[l]), float(t.lane_dx[l])), []).append(l)
        self.groups = []
        order = []
        for (nc, dx), lanes in sorted(keys.items()):
            idx = (off[lanes][:, None] + np.arange(nc)[None, :]).reshape(-1)
            self.groups.append(dict(n=nc, dx=dx, lanes=up(lanes, torch.long), idx=up(idx, torch.long), B=len(lanes)))
            order.append(idx)
        order = np.concatenate(order) if order else np.zeros(0, np.int64)
        inv = np.empty(C, dtype=np.int64)
        inv[order] = np.arange(C)
        self.unsort = up(inv, torch.long)          # position of cell c in the concatenation of the groups' outputs
        cell_lane = np.repeat(np.arange(L), n)
        self.cell_dx = up(np.asarray(t.lane_dx, dtype=np.float64)[cell_lane], torch.float32)
        self.T = t.T

    def rollout(self, action, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, differentiable=True,
                n_steps=None):
        """action [A] (device, float32) -> (reward, queue [T][L]).  differentiable=False: an evaluation episode (hard signals,
        hard static test; ItscpEnv.step(action, False))."""
        t, dev = self.t, self.device
        L, C = t.n_lanes, t.n_cells
        T = self.T if n_steps is None else int(n_steps)
        sq, F = int(n_inter_sq), int(frames_per_phase)
        um = float(u_max)
        a = action.reshape(-1).to(torch.float32)
        n_phase = a.numel() // sq
        err = ops.new_error_record(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        r, y = torch.zeros(C, **f32), torch.zeros(C, **f32)                  # empty lanes (MacroLane.__init__)
        u, q = torch.full((C,), um, **f32), torch.full((C,), um, **f32)
        own_r, own_u = torch.zeros(L, **f32), torch.full((L,), um, **f32)     # stored downstream ghosts (no flow by default)
        one = torch.ones((), **f32)
        descs = [ops.macro_desc(g["B"], g["n"], dt, g["dx"], um) for g in self.groups]
        hist_r, hist_u = [], []
        ueq_src = (um * (1.0 - torch.sqrt(torch.clamp(self.schedule, min=0.0) + 1e-5))).to(torch.float32)      # compute_u_eq of the inflow, [T][L]
        sched32 = self.schedule.to(torch.float32)
        for step in range(T):
            # ---- signals (_env.py:885-962): within a phase the action splits it, west-east green while progress < a
            ph = min(step // F, n_phase - 1)
            progress = min((step % F) / F, 1.0)
            ap = a[ph * sq:(ph + 1) * sq]
            if differentiable:
                we, ns = _soft(ap - progress, 32.0), _soft(progress - ap, 32.0)
            else:
                we, ns = (ap > progress).to(torch.float32), (progress > ap).to(torch.float32)
            lane_sig = torch.where(self.kind == 0, one, torch.where(self.kind == 1, we[self.inter], ns[self.inter]))       # [L]
            # ---- upstream ghosts (_simulator.py:42-55): the connected upstream cell (or the inflow) under the upstream lane's signal
            src, gate = self.left_src[step], self.left_gate[step]
            has = src >= 0
            srcc = torch.clamp(src, min=0)
            g_r = torch.where(has, r[self.last[srcc]], sched32[step])
            g_u = torch.where(has, u[self.last[srcc]], ueq_src[step])
            s = torch.where(gate == -2, one, torch.where(gate == -1, torch.zeros((), **f32), lane_sig[torch.clamp(gate, min=0)]))
            lf_r = g_r * s + 0.0 * (1.0 - s)
            lf_u = g_u * s + um * (1.0 - s)
            # ---- downstream ghosts (:56-60): the connected downstream cell (or the lane's own stored ghost) under the lane's own signal
            src = self.right_src[step]
            has = src >= 0
            srcc = torch.clamp(src, min=0)
            g_r = torch.where(has, r[self.first[srcc]], own_r)
            g_u = torch.where(has, u[self.first[srcc]], own_u)
            s2 = _soft(lane_sig - 0.5, 32.0) if differentiable else (lane_sig > 0.5).to(torch.float32)
            rt_r = s2 * g_r + (1.0 - s2) * 1.0
            rt_u = s2 * g_u + (1.0 - s2) * 0.0
            own_r, own_u = rt_r, rt_u
            gy, gq = _StateFromRU.apply(torch.cat([lf_r, rt_r]), torch.cat([lf_u, rt_u]), um)
            ghost_ry = torch.stack([torch.stack([lf_r, gy[:L]], dim=-1), torch.stack([rt_r, gy[L:]], dim=-1)], dim=1)          # [L][2][2]
            ghost_uq = torch.stack([torch.stack([lf_u, gq[:L]], dim=-1), torch.stack([rt_u, gq[L:]], dim=-1)], dim=1).detach()
            # ---- one ARZ step per group of lanes
            outs = [[], [], [], []]
            for g, desc in zip(self.groups, descs):
                B, n = g["B"], g["n"]
                sel = g["idx"]
                res = _LaneBatchStep.apply(r[sel].reshape(B, n), y[sel].reshape(B, n), ghost_ry[g["lanes"]],
                                           u[sel].reshape(B, n).detach(), q[sel].reshape(B, n).detach(), ghost_uq[g["lanes"]], desc, err)
                for o, x in zip(outs, res):
                    o.append(x.reshape(-1))
            nr, ny, nu, nq = (torch.cat(o)[self.unsort] for o in outs)
            r, y = nr, ny
            u = _SpeedTap.apply(nr, ny, nu, um)
            q = nq
            hist_r.append(r)
            hist_u.append(u)
        if T == 0:
            return torch.zeros((), **f32), torch.zeros(0, L, **f32)
        R_, U_ = torch.stack(hist_r), torch.stack(hist_u)                       # [T][C]: the state after every step
        # ---- queue loss (_env.py:586-618, 705-733): sigmoid(k (s0 - u)) r dx / vehicle_length summed per lane, squared, times dt;
        #      k = 16 / |running mean of all samples (s0 - u) so far, one per cell in visiting order, over the last 100 000|
        x = float(static_speed) - U_
        if differentiable:
            with torch.no_grad():
                xs = x.reshape(-1).to(torch.float64)
                cs = torch.cumsum(xs, 0)
                idx = torch.arange(1, xs.numel() + 1, device=dev)
                cs_old = torch.where(idx > WINDOW, cs[torch.clamp(idx - WINDOW - 1, min=0)], torch.zeros((), dtype=torch.float64, device=dev))
                mean = (cs - cs_old) / torch.clamp(idx, max=WINDOW).to(torch.float64)
                k = (16.0 / mean.to(torch.float32).abs()).reshape(T, C)
            static = torch.sigmoid(torch.clamp(x * k, -16.0, 16.0))
        else:
            static = (U_ < float(static_speed)).to(torch.float32)
        contrib = static * (R_ * (self.cell_dx / float(vehicle_length)))
        queue = torch.zeros(T, L, **f32)
        for g in self.groups:
            qg = contrib[:, g["idx"]].reshape(T, g["B"], g["n"]).sum(dim=-1)
            queue = queue.index_copy(1, g["lanes"], (qg * qg) * float(dt))
        reward = -queue.sum()
        ops.raise_on_fault(err)
        return reward, queue
