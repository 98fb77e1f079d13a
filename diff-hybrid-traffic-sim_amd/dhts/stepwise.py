"""Road networks of ANY size and ANY mix of ARZ cell lanes and IDM vehicle lanes, step by step on the device (round 5).

The fused network kernels keep a replica in one workgroup (dhts_net_hybrid_rollout_*: cells + lanes <= 960, <= 64 IDM lanes,
<= 128 vehicles per episode).  The reference's environment builds any grid in any mode (example/control/itscp/_env.py:221-506:
`--lane_length`, `--n_lane`, `--n_intersection` are free); beyond those limits an episode used to run lane by lane through the
mirror classes -- one operator launch plus host work per lane and step, minutes per episode.  `StepwiseNetwork` is the tier in
between: dhts_netstep_rollout_fwd / _bwd (csrc/netstep_hybrid.hip) step the whole network with a handful of launches per step
-- boundaries (ghost cells | admission, head gaps, IDM steps), the straight-lane operator once per group of ARZ lanes, hand-offs
+ queue loss -- all T steps enqueued by one host call each way.  Semantics per step: RoadNetwork.forward
(road/network/road_network.py:79-170) under ItscpRoadNetwork's signal boundaries (example/control/itscp/_simulator.py:56-276) and
ItscpEnv's queue loss (_env.py:586-618, 664-797).

Tables: dhts.network.HybridNetworkTables (a MacroNetworkTables is converted: every lane an ARZ lane) + the pre-drawn vehicle
routes, exactly what dhts.ops.DeviceHybridTables takes.  Cells are renumbered GROUP-MAJOR on the device (lanes of equal cells
and cell length contiguous, the batched operator works on the state arrays in place); lane ids stay the network's own, because
conversions, loss samples and running means follow lane-id order.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops


def default_lane_capacity(tables, vehicle_length, ceiling=32):
    """Vehicle slots per micro lane sized from the geometry: the longest micro lane's length in vehicles + 2, as a power of two in
    4 .. `ceiling`.  Vehicles keep a gap (IDM), so a lane cannot hold more; an episode that does anyway comes back as
    DHTS_FAULT_CAPACITY (ItscpEnv retries it with more room).  Small capacities keep the persistent kernels' vehicle arrays in LDS."""
    t = as_hybrid_tables(tables)
    micro = np.asarray(t.lane_macro) == 0
    if not micro.any():
        return 4
    n = int(float(np.max(np.asarray(t.lane_length, dtype=np.float64)[micro])) / float(vehicle_length)) + 2
    cap = 4
    while cap < n:
        cap *= 2
    return min(cap, int(ceiling))


def persistent_form_pays(tables):
    """Which form a SINGLE episode of this network should take: the persistent kernels (one workgroup for the whole network) while
    its static tables fit a workgroup's LDS and a thread has a handful of items per phase -- 2.5-3 x faster than the stepwise form
    at 250-400 lanes; at 1 296 lanes / 784 IDM lanes the stepwise form, which spreads a step over the chip, is 1.4 x faster
    (tests/test_stepwise_gpu.py::test_network_wider_than_the_workgroup).  Replica batches always take the persistent form."""
    t = as_hybrid_tables(tables)
    return t.n_lanes <= 1024 and t.n_cells <= 4096


def as_hybrid_tables(t):
    """A dhts.network.MacroNetworkTables as HybridNetworkTables (every lane an ARZ lane)."""
    from .network import HybridNetworkTables
    if isinstance(t, HybridNetworkTables):
        return t
    h = HybridNetworkTables.__new__(HybridNetworkTables)
    L = t.n_lanes
    h.lane_macro = np.ones(L, dtype=np.int32)
    h.lane_ncell, h.lane_off, h.n_lanes, h.n_cells, h.T = t.lane_ncell, t.lane_off, L, t.n_cells, t.T
    h.lane_dx = t.lane_dx
    h.lane_length = np.asarray(t.lane_dx, dtype=np.float64) * np.asarray(t.lane_ncell)
    h.sig_kind, h.inter = t.sig_kind, t.inter
    h.left_src, h.left_gate, h.right_src, h.schedule = t.left_src, t.left_gate, t.right_src, t.schedule
    h.conv_next = -np.ones((t.T, L), dtype=np.int32)
    h.lane_source = np.zeros(L, dtype=np.int32)
    h.draws = None
    h.nxt_ptr, h.nxt_idx, h.prv_ptr, h.prv_idx, h.n_edges = t.nxt_ptr, t.nxt_idx, t.prv_ptr, t.prv_idx, t.n_edges
    nxt = [[int(b) for b in t.nxt_idx[t.nxt_ptr[a]:t.nxt_ptr[a + 1]]] for a in range(L)]
    prv = [[int(b) for b in t.prv_idx[t.prv_ptr[a]:t.prv_ptr[a + 1]]] for a in range(L)]
    h.next_lanes, h.prev_lanes = nxt, prv
    return h


class StepwiseNetwork:
    """A network's tables on the device for dhts_netstep_rollout_fwd / _bwd.  `routes` [n][stride <= 32] int (-1 padded): the k-th
    vehicle spawned onto a micro lane takes that lane's k-th route (cyclically; waiting lists of micro source lanes in admission
    order, no wrap-around) -- as for dhts.ops.DeviceHybridTables.  lane_capacity: vehicles a micro lane holds at once (1 .. 1024)."""

    def __init__(self, tables, routes, device, lane_capacity=32, max_events=0, persistent=False, vehicle_params=None):
        """persistent: the whole episode in ONE kernel per direction, one workgroup per replica (csrc/netstep_hybrid.hip: the same
        device functions with workgroup barriers instead of kernel boundaries; ~10 x fewer microseconds per step for the grids the
        reference builds).  `tables` may then be a LIST of tables of one topology (own schedules / per-step routes / draws each): one
        replica per entry, rollout() takes action [R][A] and returns [R]-shaped results."""
        from .network import group_routes
        many = isinstance(tables, (list, tuple))
        tabs = [as_hybrid_tables(x) for x in tables] if many else [as_hybrid_tables(tables)]
        if many and not persistent:
            raise ValueError("StepwiseNetwork: a list of tables (replicas) needs persistent=True")
        t = tabs[0]
        for i, x in enumerate(tabs):
            if (x.n_lanes, x.n_cells, x.T) != (t.n_lanes, t.n_cells, t.T) or not np.array_equal(x.lane_macro, t.lane_macro) or \
                    not np.array_equal(x.lane_ncell, t.lane_ncell) or not np.array_equal(np.asarray(x.lane_source), np.asarray(t.lane_source)):
                raise ValueError("per-replica tables must share the topology of table 0: table %d differs" % i)
        self.n_replicas = len(tabs) if many else 0            # 0: one network, un-batched shapes
        self.tabs = tabs
        self.persistent = bool(persistent)
        self.t, self.device = t, device
        L, T = t.n_lanes, t.T
        self.n_lanes, self.n_cells, self.T = L, t.n_cells, T
        if not (1 <= int(lane_capacity) <= 1024):
            raise ValueError("lane_capacity must be 1 .. 1024")
        self.lane_capacity, self.max_events = int(lane_capacity), int(max_events)
        lane_macro = np.asarray(t.lane_macro, dtype=np.int32)
        ncell = np.asarray(t.lane_ncell, dtype=np.int64)
        if np.asarray(t.lane_source).any() and any(getattr(x, "draws", None) is None for x in tabs):
            raise ValueError("a network with micro source lanes needs its admission draws (HybridNetworkTables.set_micro_sources)")
        # ---- groups of ARZ lanes with equal (cells, cell length): contiguous blocks of the group-major cell / lane order ----
        keys = {}
        for l in range(L):
            if lane_macro[l]:
                keys.setdefault((int(ncell[l]), float(t.lane_dx[l])), []).append(l)
        self.groups = []
        lane_off = np.zeros(L, dtype=np.int32)
        lane_gpos = -np.ones(L, dtype=np.int32)
        pos = cell = 0
        for (nc, dx), lanes in sorted(keys.items()):
            self.groups.append((pos, len(lanes), nc, cell, dx))
            for i, l in enumerate(lanes):
                lane_off[l] = cell + i * nc
                lane_gpos[l] = pos + i
            pos += len(lanes)
            cell += len(lanes) * nc
        assert cell == t.n_cells
        self.lane_off_dev_order = lane_off
        # cell c of the network's own (lane-major) order sits at cell_pos[c] of the device order
        own_off = np.asarray(t.lane_off, dtype=np.int64)
        self.cell_pos = np.concatenate([lane_off[l] + np.arange(ncell[l]) for l in range(L) if lane_macro[l]] or [np.zeros(0, np.int64)]).astype(np.int64)
        order = np.concatenate([own_off[l] + np.arange(ncell[l]) for l in range(L) if lane_macro[l]] or [np.zeros(0, np.int64)])
        assert np.array_equal(order, np.arange(t.n_cells)), "lane_off of the tables must be lane-major over the macro lanes"
        self._garr = (_lib.NetstepGroup * max(1, len(self.groups)))(*[_lib.NetstepGroup(*g) for g in self.groups])
        # the persistent kernels' maps: interface item lane_off[l] + lane_gpos[l] + k (k = 0 .. n) -> lane; cell -> lane
        if_lane = np.zeros(max(1, t.n_cells + pos), dtype=np.int32)
        cell_lane = np.zeros(max(1, t.n_cells), dtype=np.int32)
        for l in range(L):
            if lane_macro[l]:
                b = lane_off[l] + lane_gpos[l]
                if_lane[b:b + ncell[l] + 1] = l
                cell_lane[lane_off[l]:lane_off[l] + ncell[l]] = l
        # ---- micro lanes, flux capacitors ----
        micro = [l for l in range(L) if not lane_macro[l]]
        mslot = -np.ones(L, dtype=np.int32)
        mslot[micro] = np.arange(len(micro))
        caps = [l for l in range(L) if lane_macro[l] and any(not lane_macro[b] for b in t.next_lanes[l])]
        cslot = -np.ones(L, dtype=np.int32)
        cslot[caps] = np.arange(len(caps))
        self.n_micro, self.n_caps = len(micro), len(caps)
        # ---- the ghost slots of every intersection (action partials are summed per intersection, in slot order) ----
        # a lane's upstream ghost looks at the signal of its gate lane: that lane has to belong to the same intersection
        for l in range(L):
            if lane_macro[l]:
                for g_ in t.prev_lanes[l]:
                    if t.sig_kind[g_] != 0 and t.inter[g_] != t.inter[l]:
                        raise ValueError("lane %d is gated by the signal of lane %d, which belongs to another intersection" % (l, g_))
        self._csr_sq = -1
        self._inter = np.asarray(t.inter, dtype=np.int64)
        # ---- routes ----
        routes = np.ascontiguousarray(routes, dtype=np.int32)
        if routes.ndim != 2 or routes.shape[0] < 1 or routes.shape[1] > 32:
            raise ValueError("routes must be [n_routes >= 1][stride <= 32]")
        vp = None
        if vehicle_params is not None:        # [n_routes][6], rows as `routes` (dhts_hybrid_tables::veh_params)
            routes, route_ptr, vp = group_routes(routes, L, vehicle_params)
        else:
            routes, route_ptr = group_routes(routes, L)
        self.n_routes, self.route_stride = int(routes.shape[0]), int(routes.shape[1])
        up = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)      # noqa: E731
        pad1 = lambda a: a if len(a) else np.zeros(1, dtype=np.int32)      # noqa: E731
        i32, f64 = torch.int32, torch.float64
        stack = (lambda name: np.stack([getattr(x, name) for x in tabs])) if many else (lambda name: getattr(t, name))
        self.d = dict(
            lane_ncell=up(t.lane_ncell, i32), lane_off=up(lane_off, i32), sig_kind=up(t.sig_kind, i32), inter=up(t.inter, i32),
            lane_dx=up(t.lane_dx, f64), left_src=up(stack("left_src"), i32), left_gate=up(stack("left_gate"), i32),
            right_src=up(stack("right_src"), i32), schedule=up(stack("schedule"), f64), nxt_ptr=up(t.nxt_ptr, i32),
            nxt_idx=up(pad1(t.nxt_idx), i32), prv_ptr=up(t.prv_ptr, i32),
            prv_idx=up(pad1(t.prv_idx), i32), lane_macro=up(lane_macro, i32), lane_len=up(t.lane_length, f64),
            conv_next=up(stack("conv_next"), i32), routes=up(routes, i32), route_ptr=up(route_ptr, i32), lane_gpos=up(lane_gpos, i32),
            if_lane=up(if_lane, i32), cell_lane=up(cell_lane, i32),
            micro_lanes=up(pad1(np.asarray(micro, dtype=np.int32)), i32), lane_mslot=up(mslot, i32),
            cap_lanes=up(pad1(np.asarray(caps, dtype=np.int32)), i32), lane_cslot=up(cslot, i32))
        self.has_sources = bool(np.asarray(t.lane_source).any())
        self.micro_tensor_ladder = bool(getattr(t, "micro_tensor_ladder", False))
        self.n_draws, self.draws_stride = 0, 0
        if self.has_sources:
            if many:
                n = max(len(x.draws) for x in tabs)
                d = np.full((len(tabs), n), 2.0)               # (a draw of 2.0 admits nobody)
                for i, x in enumerate(tabs):
                    d[i, :len(x.draws)] = x.draws
                self.n_draws, self.draws_stride = n, n
            else:
                d = np.asarray(t.draws, dtype=np.float64)
                self.n_draws = len(d)
            self.d["lane_source"], self.d["draws"] = up(t.lane_source, i32), up(d, f64)
        if vp is not None:
            self.d["veh_params"] = up(vp, f64)
        self.err = ops.new_error_record(device)

    # ---- per-episode data in place (same topology): new schedules / per-step routes / draws ----
    def update(self, tables):
        if self.n_replicas:
            raise ValueError("StepwiseNetwork.update: build a new network for other replica tables")
        t = as_hybrid_tables(tables)
        o = self.t
        same = (t.n_lanes == o.n_lanes and t.n_cells == o.n_cells and t.T == o.T and np.array_equal(t.lane_ncell, o.lane_ncell)
                and np.array_equal(t.lane_macro, o.lane_macro) and np.array_equal(t.nxt_idx, o.nxt_idx) and np.array_equal(t.nxt_ptr, o.nxt_ptr)
                and np.array_equal(t.sig_kind, o.sig_kind) and np.array_equal(t.inter, o.inter) and np.array_equal(t.lane_dx, o.lane_dx))
        if not same:
            raise ValueError("StepwiseNetwork.update: the network's topology changed; build a new one")
        self.t = t
        # NEW tensors, not copies into the old ones: a differentiable rollout that has not run its reverse sweep yet keeps
        # references to the tables it was stepped with (several episodes summed before one backward(): Trainer.train_epoch with
        # num_episode_per_epoch > 1)
        for name, dt in (("left_src", torch.int32), ("left_gate", torch.int32), ("right_src", torch.int32), ("conv_next", torch.int32),
                         ("schedule", torch.float64)):
            self.d[name] = torch.as_tensor(np.ascontiguousarray(getattr(t, name)), dtype=dt, device=self.device)

    def set_draws(self, draws):
        if not self.has_sources:
            raise ValueError("the network has no micro source lanes")
        d = torch.as_tensor(np.ascontiguousarray(draws, dtype=np.float64))
        if d.shape != self.d["draws"].shape:
            raise ValueError("draws must keep their shape %s" % (tuple(self.d["draws"].shape),))
        self.d["draws"] = d.to(self.device)            # (a new tensor: see update())

    def _c(self, n_inter_sq, loss_steps):
        d = self.d
        sq = int(n_inter_sq)
        if self._csr_sq != sq:
            if self.n_lanes and int(self._inter.max()) >= sq:
                raise ValueError("the tables name intersection %d but n_inter_sq = %d" % (int(self._inter.max()), sq))
            slots = [[] for _ in range(sq)]
            for l in range(self.n_lanes):
                if self.t.lane_macro[l]:
                    slots[int(self._inter[l])] += [2 * l, 2 * l + 1]
            up = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.int32, device=self.device)      # noqa: E731
            d["inter_ptr"] = up(np.concatenate([[0], np.cumsum([len(x) for x in slots])]))
            d["inter_idx"] = up(np.array([j for x in slots for j in x] or [0]))
            self._csr_sq = sq
        p = lambda k: d[k].data_ptr()      # noqa: E731
        net = _lib.NetTables(p("lane_ncell"), p("lane_off"), p("sig_kind"), p("inter"), p("lane_dx"), p("left_src"), p("left_gate"),
                             p("right_src"), p("schedule"), self.T * self.n_lanes if self.n_replicas else 0, p("nxt_ptr"), p("nxt_idx"),
                             p("prv_ptr"), p("prv_idx"), self.t.n_edges)
        src = (p("lane_source"), p("draws")) if self.has_sources else (None, None)
        hyb = _lib.HybridTables(net, p("lane_macro"), p("lane_len"), p("conv_next"), p("routes"), p("route_ptr"), self.n_routes,
                                self.route_stride, 0, int(loss_steps), self.n_micro, src[0], src[1], self.n_draws, self.draws_stride,
                                self.lane_capacity, 1 if self.micro_tensor_ladder else 0, p("veh_params") if "veh_params" in d else None, 0)
        return _lib.NetstepTables(hyb, p("lane_gpos"), self._garr, len(self.groups), p("micro_lanes"), p("lane_mslot"), p("cap_lanes"),
                                  p("lane_cslot"), self.n_caps, p("inter_ptr"), p("inter_idx"), self.max_events, p("if_lane"), p("cell_lane"),
                                  1 if self.persistent else 0, int(d["inter_idx"].numel()))

    def rollout(self, action, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, differentiable=True,
                loss_steps=0, check_faults=True):
        """action [A] (device, float32) -> (reward_cut, reward, queue [T][L], counts [4]).  reward_cut (the reward restricted to its
        first loss_steps steps; = reward for loss_steps <= 0) is differentiable w.r.t. `action`.  differentiable=False: an
        EVALUATION episode (ItscpEnv.step(action, False): hard thresholds), nothing kept for a reverse sweep."""
        return _NetstepRollout.apply(action, self, int(n_inter_sq), int(frames_per_phase), float(dt), float(u_max), float(static_speed),
                                     float(vehicle_length), bool(differentiable), int(loss_steps), bool(check_faults))

    def hist_in_network_order(self, hist):
        """hist [T + 1][4][C] of the last rollout with the cells back in the network's own (lane-major) order."""
        idx = torch.as_tensor(self.cell_pos, dtype=torch.long, device=hist.device)
        return hist[:, :, idx]


class _NetstepRollout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, action, net, sq, F, dt, um, s0, vlen, differentiable, loss_steps, check_faults):
        R = net.n_replicas
        a = ops._f32c(action.detach().reshape(R, -1) if R else action.detach().reshape(-1), "action")
        if R and a.shape[0] != R:
            raise ValueError("action must be [%d][A] for this network's %d replicas" % (R, R))
        dev = a.device
        lib = _lib.lib()
        Rn = max(R, 1)
        d = _lib.NetDesc(Rn, net.n_lanes, net.n_cells, net.T, sq, F, a.numel() // Rn, dt, um, s0, vlen)
        tc = net._c(sq, loss_steps)
        ws_n = lib.dhts_netstep_workspace_bytes(C.byref(d), C.byref(tc))
        if ws_n == 0:
            raise ValueError("dhts_netstep: unsupported network / sizes")
        # the workspace holds what the reverse sweep reads (records, tape, histories): a differentiable rollout owns its own, so that
        # several episodes can be rolled out before one backward(); evaluation episodes share the network's
        if differentiable:
            ws = torch.empty(ws_n, dtype=torch.uint8, device=dev)
        else:
            ws = getattr(net, "_ws", None)
            if ws is None or ws.numel() < ws_n:
                ws = net._ws = torch.empty(ws_n, dtype=torch.uint8, device=dev)
        hist = torch.empty((Rn, net.T + 1, 4, max(net.n_cells, 1)), dtype=torch.float32, device=dev)
        queue = torch.empty(Rn, net.T, net.n_lanes, dtype=torch.float32, device=dev)
        reward = torch.empty(Rn, 2, dtype=torch.float32, device=dev)
        counts = torch.zeros(Rn, 4, dtype=torch.int32, device=dev)
        net.err.zero_()
        _lib.check(lib.dhts_netstep_rollout_fwd(C.byref(d), C.byref(tc), 0 if differentiable else 1, ops._ptr(a), ops._ptr(hist), ops._ptr(queue),
                                                ops._ptr(reward), ops._ptr(counts), ops._ptr(ws), ops._ptr(net.err), ops._stream()),
                   "dhts_netstep_rollout_fwd")
        if check_faults:
            ops.raise_on_fault(net.err)
        net.last_hist = hist if R else hist[0]
        ctx.net, ctx.d, ctx.tc, ctx.differentiable, ctx.check_faults = net, d, tc, differentiable, check_faults
        ctx.tables = list(net.d.values())             # (tc holds raw pointers: keep the tensors of THIS episode alive, see update())
        ctx.save_for_backward(a, hist, queue.reshape(Rn, net.T, net.n_lanes), ws)
        ctx.shape = action.shape
        if not R:
            queue, counts = queue[0], counts[0]
        ctx.mark_non_differentiable(queue, counts)
        if R:
            return reward[:, 1].clone(), reward[:, 0].clone().detach(), queue, counts
        return reward[0, 1].clone(), reward[0, 0].clone().detach(), queue, counts

    @staticmethod
    def backward(ctx, g_cut, _g_reward, _g_queue, _g_counts):
        if not ctx.differentiable:
            raise RuntimeError("an evaluation episode (differentiable=False) keeps nothing for a reverse sweep")
        a, hist, queue, ws = ctx.saved_tensors
        net = ctx.net
        g = g_cut.reshape(-1).contiguous().float()
        g_action = torch.empty_like(a)
        _lib.check(_lib.lib().dhts_netstep_rollout_bwd(C.byref(ctx.d), C.byref(ctx.tc), ops._ptr(a), ops._ptr(hist), ops._ptr(queue), ops._ptr(g),
                                                       ops._ptr(g_action), ops._ptr(ws), ops._ptr(net.err), ops._stream()),
                   "dhts_netstep_rollout_bwd")
        if ctx.check_faults:
            ops.raise_on_fault(net.err)
        return (g_action.reshape(ctx.shape),) + (None,) * 10
