"""Macro road networks of ANY size, one operator call per step for all lanes (round 4).

The fused network kernels (dhts_net_macro_rollout_*) keep a replica in ONE workgroup: cells + lanes <= 1024.  The reference
builds grids of any size (example/control/itscp/_env.py:221-439: `--n_intersection=3 --n_lane=3` has 360 lanes and ~2 100
cells); until this round such a network ran lane by lane -- one dhts_macro_step_fwd launch plus host glue per lane and step,
~50 s per differentiable episode.  This module is the tier between the two: the network's lanes are the BATCH of the
straight-lane step operator (lanes grouped by (cells, cell length); dhts_macro_step_fwd / _bwd, the same HIP kernels and tape as
dMacroForwardLayer), the ghost exchange of a step (road_network.py:79-111, _simulator.py:42-60) is one kernel over all lanes
(dhts_net_ghosts_fwd; adjoint dhts_net_ghosts_bwd: csrc/netstep_kernels.hip, the ghost phases of the fused network kernels as
launches of their own), and the queue loss with its RunningMean(100 000) (_env.py:586-618, rms.py) is evaluated for all steps at
once from the state history (a prefix sum over the sample stream).  A step is ONE autograd node (`_NetStep`: three launches
forward, five backward, state and cotangents in flat device arrays whose group blocks the operators address in place);
`rollout_torch` is the same episode with the ghost exchange written as torch gathers and blends (the reference's glue arithmetic
op by op; ~40 x the launches), kept as the cross-check of the kernels.  No per-lane Python, no host round trip inside the
episode (faults are read once at the end).  The tables are dhts.network.MacroNetworkTables -- the same the fused kernels take.
"""
import ctypes as C
import numpy as np
import torch

from . import _lib, ops

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


def _addr(t, off_elems=0):
    return C.c_void_p(t.data_ptr() + 4 * int(off_elems))


class _NetStep(torch.autograd.Function):
    """One step of the whole network: (r, y [C]; own [L][2]; action [A]) -> (nr, ny, own', u', u_eq').  u, q [C] = u(r, y), u_eq(r)
    as the previous step's operator left them (values; their dependence on (r, y) is inside the operator's tape and, for the edge
    cells the ghosts read, inside dhts_net_ghosts_bwd).  Cells and lanes are in the network's GROUP-MAJOR order: a group's lanes
    and cells are contiguous, the operators work on the flat arrays in place."""

    @staticmethod
    def forward(ctx, r, y, own, action, u, q, net, step, hard):
        lib, st = _lib.lib(), ops._stream()
        r, y, own, action, u, q = (x.contiguous() for x in (r, y, own, action, u, q))
        L = net.L
        ghost = torch.empty(L, 2, 4, dtype=torch.float32, device=r.device)
        own_out = own.clone()           # (side-0 rows are not written)
        _lib.check(lib.dhts_net_ghosts_fwd(C.byref(net.desc), C.byref(net.dtab.c), int(step), int(hard), _addr(action), _addr(r), _addr(u),
                                           _addr(own), _addr(own_out), _addr(ghost), st), "dhts_net_ghosts_fwd")
        nr, ny, nu, nq = (torch.empty_like(r) for _ in range(4))
        tapes = []
        for g in net.groups:
            tape = None if hard else torch.empty(g["tape_numel"], dtype=torch.float32, device=r.device)
            co, lo = g["cell_off"], g["lane_off"]
            _lib.check(lib.dhts_macro_step_fwd(C.byref(g["desc"]), _addr(r, co), _addr(y, co), _addr(u, co), _addr(q, co), _addr(ghost, 8 * lo),
                                               _addr(nr, co), _addr(ny, co), _addr(nu, co), _addr(nq, co),
                                               _addr(tape) if tape is not None else None, _addr(net.err), st), "dhts_macro_step_fwd")
            tapes.append(tape)
        # what backward reads is snapshotted here: rollout() rebuilds the descriptors and the fault record on every call, and
        # update() overwrites the per-step tables in place (a pending backward across an update() is refused below)
        ctx.net, ctx.step, ctx.tapes = net, int(step), tapes
        ctx.desc, ctx.gdescs, ctx.err, ctx.inter, ctx.version = net.desc, [g["desc"] for g in net.groups], net.err, (net.inter_ptr, net.inter_idx), net._version
        ctx.save_for_backward(r, y, u, own, action)
        ctx.mark_non_differentiable(nu, nq)
        return nr, ny, own_out, nu, nq

    @staticmethod
    def backward(ctx, g_nr, g_ny, g_own, _g_nu, _g_nq):
        net = ctx.net
        if ctx.tapes and ctx.tapes[0] is None:
            raise RuntimeError("BatchedMacroNetwork: an evaluation episode (differentiable=False) keeps no tape to back-propagate through")
        if net._version != ctx.version:
            raise RuntimeError("BatchedMacroNetwork.update() overwrote the per-step tables this episode ran on: call backward() before update()")
        lib, st = _lib.lib(), ops._stream()
        r, y, u, own, action = ctx.saved_tensors
        g_nr, g_ny, g_own = g_nr.contiguous(), g_ny.contiguous(), g_own.contiguous()
        g_r, g_y = torch.empty_like(r), torch.empty_like(r)
        g_ghost = torch.zeros(net.L, 2, 2, dtype=torch.float64, device=r.device)
        for g, gdesc, tape in zip(net.groups, ctx.gdescs, ctx.tapes):
            co, lo = g["cell_off"], g["lane_off"]
            _lib.check(lib.dhts_macro_step_bwd(C.byref(gdesc), _addr(tape), _addr(g_nr, co), _addr(g_ny, co), _addr(g_r, co), _addr(g_y, co),
                                               C.c_void_p(g_ghost.data_ptr() + 8 * 4 * lo), _addr(ctx.err), st), "dhts_macro_step_bwd")
        g_own_out = torch.empty_like(own)
        g_action = torch.zeros_like(action)
        scratch = torch.empty(net.L, 2, 4, dtype=torch.float32, device=r.device)
        _lib.check(lib.dhts_net_ghosts_bwd(C.byref(ctx.desc), C.byref(net.dtab.c), _addr(ctx.inter[0]), _addr(ctx.inter[1]), ctx.step,
                                           _addr(action), _addr(r), _addr(y), _addr(u), _addr(own), C.c_void_p(g_ghost.data_ptr()),
                                           _addr(g_own), _addr(g_own_out), _addr(g_r), _addr(g_y), _addr(g_action), _addr(scratch), st),
                   "dhts_net_ghosts_bwd")
        return g_r, g_y, g_own_out, g_action, None, None, None, None, None


def _soft(x, c):
    """dmath/operation.py:3-30: sigmoid(clamp(x * c, -16, 16))."""
    return torch.sigmoid(torch.clamp(x * c, -16.0, 16.0))


class _CapturedReward(torch.autograd.Function):
    """reward(action) of a replayed episode: the value and d reward / d action both come out of the graph."""

    @staticmethod
    def forward(ctx, action, reward, grad):
        ctx.save_for_backward(grad)
        ctx.shape = action.shape
        return reward

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (g * grad).reshape(ctx.shape), None, None


def permuted_tables(t, perm):
    """MacroNetworkTables `t` with its lanes renumbered: new lane i = old lane perm[i] (cells follow their lanes)."""
    from .network import MacroNetworkTables
    perm = np.asarray(perm, dtype=np.int64)
    L = t.n_lanes
    inv = np.empty(L, dtype=np.int64)
    inv[perm] = np.arange(L)
    p = MacroNetworkTables.__new__(MacroNetworkTables)
    p.n_lanes, p.T = L, t.T
    p.lane_ncell = np.asarray(t.lane_ncell)[perm].astype(np.int32)
    p.lane_off = np.concatenate([[0], np.cumsum(p.lane_ncell)[:-1]]).astype(np.int32)
    p.n_cells = t.n_cells
    p.lane_dx = np.asarray(t.lane_dx)[perm].astype(np.float64)
    p.sig_kind, p.inter = np.asarray(t.sig_kind)[perm].astype(np.int32), np.asarray(t.inter)[perm].astype(np.int32)
    remap = lambda a: np.where(a >= 0, inv[np.clip(a, 0, L - 1)], a).astype(np.int32)      # noqa: E731  (negative codes stay)
    p.left_src, p.left_gate, p.right_src = (np.ascontiguousarray(remap(np.asarray(x)[:, perm])) for x in (t.left_src, t.left_gate, t.right_src))
    p.schedule = np.ascontiguousarray(np.asarray(t.schedule)[:, perm])
    nxt = [[] for _ in range(L)]
    prv = [[] for _ in range(L)]
    for a in range(L):
        for e in range(t.nxt_ptr[a], t.nxt_ptr[a + 1]):
            b = int(t.nxt_idx[e])
            nxt[inv[a]].append(int(inv[b]))
            prv[inv[b]].append(int(inv[a]))
    p.nxt_ptr = np.concatenate([[0], np.cumsum([len(x) for x in nxt])]).astype(np.int32)
    p.nxt_idx = np.array([b for x in nxt for b in sorted(x)], dtype=np.int32)
    p.prv_ptr = np.concatenate([[0], np.cumsum([len(x) for x in prv])]).astype(np.int32)
    p.prv_idx = np.array([a for x in prv for a in sorted(x)], dtype=np.int32)
    p.n_edges = int(len(p.nxt_idx))
    p.is_source = np.asarray(t.is_source)[perm]
    return p


class BatchedMacroNetwork:
    """MacroNetworkTables on the device, lanes grouped for the batched step operator.  `rollout` = ItscpEnv.step of a `macro`
    network of any size: (reward, queue [T][L])."""

    def __init__(self, tables, device):
        t = tables
        self.t, self.device = t, device
        L, Cn = t.n_lanes, t.n_cells
        self.L, self.T = L, t.T
        self._version = 0                       # counts update() calls: a pending backward across one is refused (_NetStep.backward)
        if L and int(np.min(t.inter)) < 0:
            raise ValueError("BatchedMacroNetwork: every lane needs an intersection index >= 0 (tables.inter)")
        up = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)      # noqa: E731
        off, n = np.asarray(t.lane_off, dtype=np.int64), np.asarray(t.lane_ncell, dtype=np.int64)
        # groups of lanes with the same number of cells and the same cell length: one operator call each per step
        keys = {}
        for l in range(L):
            keys.setdefault((int(n[l]), float(t.lane_dx[l])), []).append(l)
        self.groups = []
        order, perm = [], []
        cell_off = 0
        for (nc, dx), lanes in sorted(keys.items()):
            idx = (off[lanes][:, None] + np.arange(nc)[None, :]).reshape(-1)
            self.groups.append(dict(n=nc, dx=dx, lanes=up(lanes, torch.long), idx=up(idx, torch.long), B=len(lanes),
                                    lane_off=len(perm), cell_off=cell_off))
            order.append(idx)
            perm += lanes
            cell_off += len(idx)
        order = np.concatenate(order) if order else np.zeros(0, np.int64)
        inv = np.empty(Cn, dtype=np.int64)
        inv[order] = np.arange(Cn)
        self.unsort = up(inv, torch.long)          # position of (original) cell c in the group-major order
        self.lane_unsort = up(np.argsort(np.asarray(perm, dtype=np.int64)), torch.long)      # position of (original) lane l
        cell_lane = np.repeat(np.arange(L), n)
        self.cell_dx = up(np.asarray(t.lane_dx, dtype=np.float64)[cell_lane], torch.float32)
        # the same network with its lanes in group-major order: what the kernels see
        self.tp = permuted_tables(t, perm)
        self.dtab = ops.DeviceNetTables(self.tp, device)
        self.n_inter_tab = int(np.max(t.inter)) + 1 if L else 1
        self._csr_sq = -1
        # the action partials of a lane's two ghosts are summed under the LANE's intersection (dhts_net_ghosts_bwd: inter_idx); the
        # upstream ghost's signal is the gate lane's, so a signalled upstream lane has to belong to the same intersection (the fused
        # kernels make the same assumption and flag it at run time; every itscp grid satisfies it: an approaching lane gates the
        # intersection's own mid lanes)
        for l in range(L):
            for e in range(t.prv_ptr[l], t.prv_ptr[l + 1]):
                g_ = int(t.prv_idx[e])
                if t.sig_kind[g_] != 0 and t.inter[g_] != t.inter[l]:
                    raise ValueError("batched path: lane %d is gated by the signal of lane %d, which belongs to another intersection" % (l, g_))
        # original order, as torch index tensors: rollout_torch
        self.first, self.last = up(off, torch.long), up(off + n - 1, torch.long)
        self.left_src, self.left_gate, self.right_src = (up(x, torch.long) for x in (t.left_src, t.left_gate, t.right_src))      # [T][L]
        self.schedule = up(t.schedule, torch.float64)                                                                                # [T][L]
        self.kind, self.inter = up(t.sig_kind, torch.long), up(t.inter, torch.long)

    # ---- the queue loss of an episode from its state history (original lane / cell order) ------------------------------------
    def _queue_loss(self, R_, U_, dt, static_speed, vehicle_length, differentiable):
        """_env.py:586-618, 705-733: sigmoid(k (s0 - u)) r dx / vehicle_length summed per lane, squared, times dt;
        k = 16 / |running mean of all samples (s0 - u) so far, one per cell in visiting order, over the last 100 000|."""
        T, Cn = R_.shape
        L, dev = self.L, R_.device
        x = float(static_speed) - U_
        if differentiable:
            with torch.no_grad():
                xs = x.reshape(-1).to(torch.float64)
                cs = torch.cumsum(xs, 0)
                idx = torch.arange(1, xs.numel() + 1, device=dev)
                cs_old = torch.where(idx > WINDOW, cs[torch.clamp(idx - WINDOW - 1, min=0)], torch.zeros((), dtype=torch.float64, device=dev))
                mean = (cs - cs_old) / torch.clamp(idx, max=WINDOW).to(torch.float64)
                k = (16.0 / mean.to(torch.float32).abs()).reshape(T, Cn)
            static = torch.sigmoid(torch.clamp(x * k, -16.0, 16.0))
        else:
            static = (U_ < float(static_speed)).to(torch.float32)
        contrib = static * (R_ * (self.cell_dx / float(vehicle_length)))
        queue = torch.zeros(T, L, dtype=torch.float32, device=dev)
        for g in self.groups:
            qg = contrib[:, g["idx"]].reshape(T, g["B"], g["n"]).sum(dim=-1)
            queue = queue.index_copy(1, g["lanes"], (qg * qg) * float(dt))
        return -queue.sum(), queue

    def rollout(self, action, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, differentiable=True,
                n_steps=None, check_faults=True):
        """action [A] (device, float32) -> (reward, queue [T][L]).  differentiable=False: an evaluation episode (hard signals,
        hard static test; ItscpEnv.step(action, False)).  check_faults=False: the fault record (self.err) is not read back -- no host
        synchronisation, e.g. inside a HIP-graph capture."""
        dev = self.device
        L, Cn = self.L, self.t.n_cells
        T = self.T if n_steps is None else int(n_steps)
        um = float(u_max)
        a = action.reshape(-1).to(torch.float32).contiguous()
        if not differentiable:
            a = a.detach()                      # an evaluation episode keeps no tape: no autograd node may be recorded for it
        if int(n_inter_sq) < self.n_inter_tab:
            raise ValueError("the tables name intersection %d but n_inter_sq = %d" % (self.n_inter_tab - 1, n_inter_sq))
        if self._csr_sq != int(n_inter_sq):          # the ghost slots (2 lane + side) of every intersection, ascending
            slots = [[] for _ in range(int(n_inter_sq))]
            for l in range(L):
                slots[int(self.tp.inter[l])] += [2 * l, 2 * l + 1]
            up = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.int32, device=dev)      # noqa: E731
            self.inter_ptr = up(np.concatenate([[0], np.cumsum([len(x) for x in slots])]))
            self.inter_idx = up(np.array([j for x in slots for j in x] or [0]))
            self._csr_sq = int(n_inter_sq)
        self.desc = _lib.NetDesc(1, L, Cn, self.T, int(n_inter_sq), int(frames_per_phase), a.numel(), float(dt), um, float(static_speed),
                                 float(vehicle_length))
        self.err = ops.new_error_record(dev)
        for g in self.groups:
            g["desc"] = ops.macro_desc(g["B"], g["n"], dt, g["dx"], um)
            g["tape_numel"] = ops.macro_step_tape_numel(g["desc"])
        f32 = dict(dtype=torch.float32, device=dev)
        if T == 0:
            return torch.zeros((), **f32), torch.zeros(0, L, **f32)
        r, y = torch.zeros(Cn, **f32), torch.zeros(Cn, **f32)                 # empty lanes (MacroLane.__init__), group-major order
        u, q = torch.full((Cn,), um, **f32), torch.full((Cn,), um, **f32)
        own = torch.stack([torch.zeros(L, **f32), torch.full((L,), um, **f32)], dim=-1).contiguous()     # stored downstream ghosts: no flow
        hr, hy, hu = [], [], []
        with torch.set_grad_enabled(bool(differentiable) and torch.is_grad_enabled()):
            for step in range(T):
                r, y, own, u, q = _NetStep.apply(r, y, own, a, u, q, self, step, not differentiable)
                hr.append(r)
                hy.append(y)
                hu.append(u)
        R_, Y_, U_ = torch.stack(hr), torch.stack(hy), torch.stack(hu)        # [T][C]: the state after every step
        if differentiable:
            U_ = _SpeedTap.apply(R_.reshape(-1), Y_.reshape(-1), U_.reshape(-1), um).reshape(T, Cn)      # u = u(r, y) of FullQ.set_r_y
        reward, queue = self._queue_loss(R_[:, self.unsort], U_[:, self.unsort], dt, static_speed, vehicle_length, differentiable)
        if check_faults:
            ops.raise_on_fault(self.err)
        return reward, queue

    # ---- the per-step tables of a new episode (schedules, per-step routes) into the device tables, in place ------------------
    def update(self, tables):
        """Same topology, new schedules / per-step routes (ItscpEnv.reset draws them anew): the device tables are overwritten in
        place, so a captured episode (graphed_rollout) stays valid.  Raises ValueError when the topology differs."""
        t = self.t
        same = (tables.n_lanes == t.n_lanes and tables.n_cells == t.n_cells and tables.T == t.T and
                np.array_equal(tables.lane_ncell, t.lane_ncell) and np.array_equal(tables.nxt_idx, t.nxt_idx) and
                np.array_equal(tables.nxt_ptr, t.nxt_ptr) and np.array_equal(tables.sig_kind, t.sig_kind) and
                np.array_equal(tables.inter, t.inter) and np.allclose(tables.lane_dx, t.lane_dx, rtol=0, atol=0))
        if not same:
            raise ValueError("BatchedMacroNetwork.update: the network's topology changed; build a new one")
        self.t = tables
        self._version += 1
        order = np.argsort(self.lane_unsort.cpu().numpy())          # new lane i = old lane order[i]
        tp = permuted_tables(tables, order)
        self.tp = tp
        d = self.dtab
        for name in ("left_src", "left_gate", "right_src"):
            getattr(d, name).copy_(torch.as_tensor(np.ascontiguousarray(getattr(tp, name)), dtype=torch.int32))
            getattr(self, name).copy_(torch.as_tensor(np.ascontiguousarray(getattr(tables, name)), dtype=torch.long))
        d.schedule.copy_(torch.as_tensor(np.ascontiguousarray(tp.schedule), dtype=torch.float64))
        self.schedule.copy_(torch.as_tensor(np.ascontiguousarray(tables.schedule), dtype=torch.float64))

    # ---- a whole episode (forward, and backward for a differentiable one) as ONE HIP graph ------------------------------------
    def graphed_rollout(self, action, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, differentiable=True):
        """rollout() captured once into a HIP graph (8 launches per step and ~25 small allocations are what an eager episode
        pays for) and replayed for every later call with the same sizes: the action is copied into the graph's static input, the
        per-step tables are read from the device tables (update() overwrites them in place).  Returns (reward, queue) like rollout();
        reward is differentiable w.r.t. `action` through the captured backward pass.  Faults are read after the replay."""
        key = (int(n_inter_sq), int(frames_per_phase), float(dt), float(u_max), float(static_speed), float(vehicle_length), bool(differentiable),
               int(action.numel()))
        g = getattr(self, "_graphs", None)
        if g is None:
            g = self._graphs = {}
        if key not in g:
            static_a = action.detach().reshape(-1).to(torch.float32).clone().requires_grad_(differentiable)
            args = key[:6]

            def episode():
                if differentiable:
                    static_a.grad = None
                    with torch.enable_grad():           # (the caller may sit under no_grad: the captured episode differentiates itself)
                        reward, queue = self.rollout(static_a, *args, differentiable=True, check_faults=False)
                        (grad,) = torch.autograd.grad(reward, static_a)
                    return reward.detach(), queue.detach(), grad
                with torch.no_grad():
                    reward, queue = self.rollout(static_a, *args, differentiable=False, check_faults=False)
                return reward, queue, None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                     # warm-up outside the capture (allocator, lazy initialisations)
                for _ in range(2):
                    episode()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = episode()
            g[key] = (graph, static_a, out, self.err)
        graph, static_a, out, err = g[key]
        with torch.no_grad():
            static_a.copy_(action.detach().reshape(-1))
        err.zero_()
        graph.replay()
        ops.raise_on_fault(err)
        reward, queue, grad = out
        if differentiable and action.requires_grad:
            reward = _CapturedReward.apply(action, reward.clone(), grad.clone())
        else:
            reward = reward.clone()
        return reward, queue.clone()

    def rollout_torch(self, action, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, differentiable=True,
                      n_steps=None):
        """The same episode with the ghost exchange as torch gathers and blends (the reference's glue op by op) and autograd
        between the pieces: the cross-check of dhts_net_ghosts_fwd / _bwd (and ~40 x the launches)."""
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
        # a source lane's upstream ghost reaches the step operator in double (the reference's Python floats, _simulator.py:68-71): the
        # quad {NaN, 0, low word of r, high word of r} of csrc/arz_device.hpp::ghost_source_pack
        sched_words = self.schedule.contiguous().view(torch.int32).reshape(self.schedule.shape[0], L, 2).view(torch.float32)          # [T][L][lo, hi]
        nan32, zero32 = torch.full((), float("nan"), **f32), torch.zeros((), **f32)
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
            is_src = ~has
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
            lf_r_, lf_y_ = torch.where(is_src, nan32, lf_r), torch.where(is_src, zero32, gy[:L])
            lf_u_, lf_q_ = torch.where(is_src, sched_words[step, :, 0], lf_u), torch.where(is_src, sched_words[step, :, 1], gq[:L])
            ghost_ry = torch.stack([torch.stack([lf_r_, lf_y_], dim=-1), torch.stack([rt_r, gy[L:]], dim=-1)], dim=1)          # [L][2][2]
            ghost_uq = torch.stack([torch.stack([lf_u_, lf_q_], dim=-1), torch.stack([rt_u, gq[L:]], dim=-1)], dim=1).detach()
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
        reward, queue = self._queue_loss(torch.stack(hist_r), torch.stack(hist_u), dt, static_speed, vehicle_length, differentiable)
        ops.raise_on_fault(err)
        return reward, queue
