"""Flat tables describing a road network with signals (the itscp environment in `macro` / `hybrid` mode) for the fused
network kernels: lane geometry, per-lane signal kind, and per time step which neighbour feeds each ghost cell.

Built on the host from plain arrays (lane cell counts / lengths, directed lane edges, per-step macro-route matching,
inflow schedules); semantics follow ItscpRoadNetwork.setup_macro_boundary (reference example/control/itscp/
_simulator.py:56-137) and RoadNetwork.get_macro_boundary (road/network/road_network.py:299-362).
"""
import numpy as np

SIG_ALWAYS, SIG_WE, SIG_NS = 0, 1, 2


class MacroNetworkTables:
    def __init__(self, lane_ncell, lane_length, edges, sig_kind, inter, macro_route, schedule):
        """lane_ncell [L] int, lane_length [L] float, edges [(a, b)] lane a feeds lane b, sig_kind / inter [L] int,
        macro_route [T][L] int (successor chosen for lane l at step t, -1 = none), schedule [L][T] float."""
        self.lane_ncell = np.asarray(lane_ncell, dtype=np.int32)
        L = len(self.lane_ncell)
        self.n_lanes = L
        self.lane_off = np.concatenate([[0], np.cumsum(self.lane_ncell)[:-1]]).astype(np.int32)
        self.n_cells = int(self.lane_ncell.sum())
        self.lane_dx = (np.asarray(lane_length, dtype=np.float64) / self.lane_ncell).astype(np.float64)
        self.sig_kind = np.asarray(sig_kind, dtype=np.int32)
        self.inter = np.asarray(inter, dtype=np.int32)
        macro_route = np.asarray(macro_route, dtype=np.int32)
        T = macro_route.shape[0]
        self.T = T
        prev = [[] for _ in range(L)]
        nxt = [[] for _ in range(L)]
        for a, b in np.asarray(edges, dtype=np.int64).tolist():
            nxt[a].append(b)
            prev[b].append(a)
        # static adjacency (CSR, ascending ids)
        self.nxt_ptr = np.concatenate([[0], np.cumsum([len(n) for n in nxt])]).astype(np.int32)
        self.nxt_idx = np.array([b for n in nxt for b in sorted(n)], dtype=np.int32)
        self.prv_ptr = np.concatenate([[0], np.cumsum([len(p_) for p_ in prev])]).astype(np.int32)
        self.prv_idx = np.array([a for p_ in prev for a in sorted(p_)], dtype=np.int32)
        self.n_edges = int(len(self.nxt_idx))
        if max([len(n) for n in nxt] + [len(p_) for p_ in prev]) > 4:
            raise ValueError("the network kernels support at most 4 upstream and 4 downstream lanes per lane")
        route_prev = -np.ones((T, L), dtype=np.int32)
        tt, aa = np.nonzero(macro_route >= 0)
        route_prev[tt, macro_route[tt, aa]] = aa
        self.left_src = np.empty((T, L), dtype=np.int32)
        self.left_gate = np.empty((T, L), dtype=np.int32)
        self.right_src = np.empty((T, L), dtype=np.int32)
        for l in range(L):
            if len(prev[l]) == 0:
                self.left_src[:, l] = -1                 # source lane: inflow schedule, always open
                self.left_gate[:, l] = -2
            else:
                self.left_src[:, l] = prev[l][0] if len(prev[l]) == 1 else route_prev[:, l]
                self.left_gate[:, l] = route_prev[:, l]  # -1: nobody routed into this lane at this step = red
            if len(nxt[l]) == 0:
                self.right_src[:, l] = -1                # sink lane: its own stored ghost
            else:
                self.right_src[:, l] = nxt[l][0] if len(nxt[l]) == 1 else macro_route[:, l]
        assert (self.left_src[:, [len(p) > 0 for p in prev]] >= 0).all(), "a lane with several upstream lanes was left unmatched"
        assert (self.right_src[:, [len(n) > 0 for n in nxt]] >= 0).all(), "a lane with several downstream lanes was left unmatched"
        self.schedule = np.ascontiguousarray(np.asarray(schedule, dtype=np.float64).T)      # [T][L]
        self.is_source = np.array([len(p) == 0 for p in prev])

    @staticmethod
    def from_env(env):
        """From an example.control.itscp._env.ItscpEnv (after reset) in `macro` mode."""
        keys = list(env.lane.keys())
        lanes = [env.lane[k].sim_lane for k in keys]
        assert all(sl.is_macro() for sl in lanes), "macro mode only"
        kinds = []
        for k in keys:
            if k.loc == "mid" or not k.approaching:
                kinds.append(SIG_ALWAYS)
            else:
                kinds.append(SIG_WE if k.loc in ("west", "east") else SIG_NS)
        edges = [(a, b) for a in env.simulator.lane for b in env.simulator.lane[a].next_lane.keys()]
        T, L = env.num_timestep, len(keys)
        mr = -np.ones((T, L), dtype=np.int32)
        for t, r in enumerate(env.macro_route_schedule):
            for a, b in r.next_lane_dict.items():
                mr[t, a] = b
        return MacroNetworkTables([sl.num_cell for sl in lanes], [sl.length for sl in lanes], edges, kinds,
                                  [k.row * env.num_intersection + k.col for k in keys], mr,
                                  [env.schedule[k] for k in keys])


def group_routes(routes, n_lanes, vehicle_params=None):
    """Pre-drawn vehicle routes [n][stride] (-1 padded) -> (rows stably sorted by their first lane, route_ptr [L + 1]).
    A spawn order recorded from the reference stays intact per lane, which is all the hand-off logic can observe.
    vehicle_params [n][6] (the IDM attributes of the vehicle that takes each row): returned third, in the rows' new order."""
    routes = np.ascontiguousarray(routes, dtype=np.int32)
    first = routes[:, 0]
    order = np.argsort(first, kind="stable")
    ptr = np.zeros(n_lanes + 1, dtype=np.int32)
    for f in first:
        if f >= 0:
            ptr[f + 1] += 1
    if vehicle_params is not None:
        vp = np.ascontiguousarray(vehicle_params, dtype=np.float64)
        if vp.shape != (routes.shape[0], 6):
            raise ValueError("vehicle_params must be [%d routes][6] (accel_max, accel_pref, target_speed, min_space, time_pref, length)" % routes.shape[0])
        return routes[order], np.cumsum(ptr).astype(np.int32), np.ascontiguousarray(vp[order])
    return routes[order], np.cumsum(ptr).astype(np.int32)


class HybridNetworkTables:
    """Tables of a network mixing macro lanes (ARZ cells) and micro lanes (IDM vehicles): the itscp environment in
    `hybrid` mode (reference _env.py:489-498: the lanes of interior intersections are micro).  Ghost sources follow
    RoadNetwork.get_macro_boundary (road_network.py:299-362): a micro neighbour never feeds a ghost cell, the lane's
    own stored ghost does; hand-off targets follow RoadNetwork.conversion (road_network.py:113-170)."""

    def __init__(self, lane_macro, lane_ncell, lane_length, edges, sig_kind, inter, macro_route, schedule):
        self.lane_macro = np.asarray(lane_macro, dtype=np.int32)
        self.lane_ncell = np.where(self.lane_macro == 1, np.asarray(lane_ncell, dtype=np.int32), 0).astype(np.int32)
        self.lane_length = np.asarray(lane_length, dtype=np.float64)
        L = len(self.lane_ncell)
        self.n_lanes = L
        self.lane_off = np.concatenate([[0], np.cumsum(self.lane_ncell)[:-1]]).astype(np.int32)
        self.n_cells = int(self.lane_ncell.sum())
        self.lane_dx = np.where(self.lane_ncell > 0, self.lane_length / np.maximum(self.lane_ncell, 1), 0.0).astype(np.float64)
        self.sig_kind = np.asarray(sig_kind, dtype=np.int32)
        self.inter = np.asarray(inter, dtype=np.int32)
        macro_route = np.asarray(macro_route, dtype=np.int32)
        T = macro_route.shape[0]
        self.T = T
        prev = [[] for _ in range(L)]
        nxt = [[] for _ in range(L)]
        for a, b in np.asarray(edges, dtype=np.int64).tolist():
            nxt[a].append(b)
            prev[b].append(a)
        self.next_lanes, self.prev_lanes = nxt, prev
        is_macro = self.lane_macro == 1
        route_prev = -np.ones((T, L), dtype=np.int32)
        tt, aa = np.nonzero(macro_route >= 0)
        route_prev[tt, macro_route[tt, aa]] = aa
        self.left_src = -np.ones((T, L), dtype=np.int32)
        self.left_gate = -np.ones((T, L), dtype=np.int32)
        self.right_src = -np.ones((T, L), dtype=np.int32)
        for l in range(L):
            if not is_macro[l]:
                continue
            if len(prev[l]) == 0:
                self.left_src[:, l] = -1
                self.left_gate[:, l] = -2
            elif len(prev[l]) == 1:
                self.left_src[:, l] = prev[l][0] if is_macro[prev[l][0]] else -3
                self.left_gate[:, l] = route_prev[:, l]
            else:
                if not all(is_macro[a] for a in prev[l]):
                    raise ValueError("a macro lane with several upstream lanes needs all of them macro")
                self.left_src[:, l] = route_prev[:, l]
                self.left_gate[:, l] = route_prev[:, l]
                if (route_prev[:, l] < 0).any():
                    raise ValueError("a lane with several upstream lanes was left unmatched")
            if len(nxt[l]) == 1:
                self.right_src[:, l] = nxt[l][0] if is_macro[nxt[l][0]] else -1
            elif len(nxt[l]) > 1:
                if not all(is_macro[b] for b in nxt[l]):
                    raise ValueError("a macro lane with several downstream lanes needs all of them macro")
                self.right_src[:, l] = macro_route[:, l]
                if (macro_route[:, l] < 0).any():
                    raise ValueError("a lane with several downstream lanes was left unmatched")
        # micro SOURCE lanes (no upstream lane): they admit waiting vehicles stochastically (_simulator.py:153-174).  The host's
        # admission draws are data for the kernels: `draws` (set_micro_sources) is the stream np.random.random() would yield, in
        # call order; the k-th vehicle admitted to such a lane takes the lane's k-th route row (callers list the waiting routes
        # in admission order, i.e. the reference's waiting list reversed -- it pops from the end)
        self.lane_source = np.array([int((not is_macro[l]) and len(prev[l]) == 0) for l in range(L)], dtype=np.int32)
        self.draws = None
        self.conv_next = np.where(is_macro[None, :], macro_route, -1).astype(np.int32)
        # static adjacency (CSR, ascending ids) for the reverse sweep's inbox routing
        self.nxt_ptr = np.concatenate([[0], np.cumsum([len(n) for n in nxt])]).astype(np.int32)
        self.nxt_idx = np.array([b for n in nxt for b in sorted(n)], dtype=np.int32)
        self.prv_ptr = np.concatenate([[0], np.cumsum([len(p_) for p_ in prev])]).astype(np.int32)
        self.prv_idx = np.array([a for p_ in prev for a in sorted(p_)], dtype=np.int32)
        self.n_edges = int(len(self.nxt_idx))
        if max([len(n) for n in nxt] + [len(p_) for p_ in prev]) > 4:
            raise ValueError("the network kernels support at most 4 upstream and 4 downstream lanes per lane")
        self.schedule = np.ascontiguousarray(np.asarray(schedule, dtype=np.float64).T)      # [T][L]

    @staticmethod
    def plain(lane_macro, lane_ncell, lane_length, edges, T, macro_route=None):
        """Tables of a PLAIN road network -- RoadNetwork of the reference without the itscp layer (example/inverse/hybrid.py): no
        signals, no inflow schedules.  A ghost is the connected macro lane's edge cell (the single neighbour, or macro_route's
        choice among several: [L] ints, successor of each macro lane, -1 = none, as RoadNetwork.create_random_macro_route draws it
        once) or the lane's own stored ghost (get_macro_boundary, road_network.py:299-362).  For
        dhts.ops.net_hybrid_state_rollout (plain = True)."""
        lane_macro = np.asarray(lane_macro, dtype=np.int32)
        L = len(lane_macro)
        mr = -np.ones(L, dtype=np.int32) if macro_route is None else np.asarray(macro_route, dtype=np.int32)
        nxt = [[] for _ in range(L)]
        for a, b in np.asarray(edges, dtype=np.int64).reshape(-1, 2).tolist():
            nxt[a].append(b)
        for l in range(L):                       # a single macro successor needs no route entry
            if lane_macro[l] == 1 and mr[l] < 0 and len(nxt[l]) == 1:
                mr[l] = nxt[l][0]
        t = HybridNetworkTables(lane_macro, lane_ncell, lane_length, edges, np.zeros(L, dtype=np.int32), np.zeros(L, dtype=np.int32),
                                np.tile(mr[None, :], (int(T), 1)), np.zeros((L, int(T))))
        # no inflow schedules: a lane without an upstream lane keeps its stored ghost (code -3), like one behind a micro lane
        t.left_src = np.where(t.left_src == -1, -3, t.left_src).astype(np.int32)
        t.is_plain = True
        return t

    def set_micro_sources(self, draws, tensor_ladder=True):
        """The admission draws of the micro source lanes: the values np.random.random() yields (or yielded, for a replay), in
        call order -- one per source lane and step in which the lane has room for a vehicle.  tensor_ladder: the lanes are itscp
        `micro` mode's plain MicroLane objects on float32 tensors (the only network the reference builds with source lanes,
        _env.py:484-498); False for a network of dMicroLane lanes that happens to have an IDM source lane."""
        self.draws = np.ascontiguousarray(draws, dtype=np.float64)
        self.micro_tensor_ladder = bool(tensor_ladder)
        return self

    def check_kernel_limits(self):
        """Raise ValueError when the network is outside what the fused hybrid kernels hold in one workgroup."""
        if self.lane_source.any() and self.draws is None:
            raise ValueError("hybrid kernels: a network with micro source lanes needs its admission draws (set_micro_sources)")
        n_micro = int((self.lane_macro == 0).sum())
        n_spawn = sum(1 for l in range(self.n_lanes)
                      if self.lane_macro[l] == 1 and any(self.lane_macro[b] == 0 for b in self.next_lanes[l]))
        if self.n_cells + self.n_lanes > 960:
            raise ValueError("hybrid kernels: cells + lanes must be <= 960 (got %d)" % (self.n_cells + self.n_lanes))
        if n_micro > 64:
            raise ValueError("hybrid kernels: at most 64 micro lanes (got %d)" % n_micro)
        if n_spawn > 16:
            raise ValueError("hybrid kernels: at most 16 macro lanes feeding micro lanes (got %d)" % n_spawn)
        if self.n_cells < 1 and n_micro < 1:
            raise ValueError("hybrid kernels: the network has neither cells nor micro lanes")

    @staticmethod
    def from_env(env):
        """From an example.control.itscp._env.ItscpEnv (after reset) in `hybrid` mode."""
        keys = list(env.lane.keys())
        lanes = [env.lane[k].sim_lane for k in keys]
        kinds = []
        for k in keys:
            if k.loc == "mid" or not k.approaching:
                kinds.append(SIG_ALWAYS)
            else:
                kinds.append(SIG_WE if k.loc in ("west", "east") else SIG_NS)
        edges = [(a, b) for a in env.simulator.lane for b in env.simulator.lane[a].next_lane.keys()]
        T, L = env.num_timestep, len(keys)
        mr = -np.ones((T, L), dtype=np.int32)
        for t, r in enumerate(env.macro_route_schedule):
            for a, b in r.next_lane_dict.items():
                mr[t, a] = b
        return HybridNetworkTables([int(sl.is_macro()) for sl in lanes], [getattr(sl, "num_cell", 0) for sl in lanes],
                                   [sl.length for sl in lanes], edges, kinds,
                                   [k.row * env.num_intersection + k.col for k in keys], mr, [env.schedule[k] for k in keys])
