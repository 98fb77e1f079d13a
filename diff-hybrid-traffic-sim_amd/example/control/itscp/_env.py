"""ItscpEnv on the reference's import path (example.control.itscp._env; reference _env.py:24-962) without the
highway-env / gym / pygame dependencies: grid-of-intersections topology, inflow schedules, differentiable signal phases,
per-step queue-length loss.  Rendering is out of scope.

Topology (reference _env.py:221-439): every intersection (row, col) has, per side (south / west / north / east) and per
lane index, one approaching and one leaving lane of `lane_length`; inside the intersection each approaching lane gets a
straight connector to the opposite leaving lane and (right-most lane only) a right-turn connector; left turns are disabled.
Neighbouring intersections are joined leaving -> approaching.  Lane geometry is only used for lane LENGTHS; it is
reproduced with the same floating-point construction so that ceil(length / cell_length) gives the same cell counts.
"""
import numpy as np
import torch as th

from dmath.operation import sigmoid
from example.common.rms import RunningMean
from example.control.itscp._env_config import default_config
from example.control.itscp._simulator import ItscpRoadNetwork as SimRoadNetwork
from road.lane.dmacro_lane import MacroLane, dMacroLane
from road.lane.dmicro_lane import MicroLane, dMicroLane

LANE_WIDTH = 4          # highway-env AbstractLane.DEFAULT_WIDTH


class LaneID:
    """(row, col) of the intersection, side `loc` ('south' / 'west' / 'north' / 'east' / 'mid'), for connectors the side
    they come from (`ploc`), direction and lane index (reference _env.py:24-60)."""

    def __init__(self, row, col, loc, ploc, approaching, lane_id):
        self.row, self.col, self.loc, self.ploc, self.approaching, self.lane_id = row, col, loc, ploc, approaching, lane_id

    def __str__(self):
        return "{}_{}_{}_{}_{}_{}".format(self.row, self.col, self.loc, self.ploc,
                                          "approaching" if self.approaching else "leaving", self.lane_id)

    def __eq__(self, o):
        return str(self) == str(o)

    def __hash__(self):
        return hash(str(self))


class _Segment:
    def __init__(self, start, end):
        self.start, self.end = np.array(start, dtype=float), np.array(end, dtype=float)
        self.length = float(np.linalg.norm(self.end - self.start))
        self.direction = (self.end - self.start) / self.length

    def position(self, s):
        return self.start + s * self.direction


class _Lane:
    def __init__(self, seg, sim_lane):
        self.env_lane, self.sim_lane = seg, sim_lane


def itscp_random_schedule(lane_id, num_timestep):
    """Five sessions of constant random inflow per lane (reference _env.py:62-92)."""
    per = num_timestep // 5
    schedule = {}
    for id in lane_id:
        cur = []
        for _ in range(5):
            r = np.random.random((1)).item()
            cur.extend([r] * per)
            cur = cur[:num_timestep]
        schedule[id] = cur
    return schedule


SIDE_OF_CORNER = (("south", "east"), ("west", "south"), ("north", "west"), ("east", "north"))   # (approaching, leaving)
STRAIGHT = {"north": "south", "west": "east", "east": "west", "south": "north"}
RIGHT = {"north": "west", "west": "south", "east": "north", "south": "east"}


class Box:
    """The three attributes of gym.spaces.Box the trainer reads (reference _env.py:170-173, trainer.py:26-36,182-187)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
        self.dtype = np.dtype(dtype)


def _vehicle_attributes(v, sim):
    """(accel_max, accel_pref, target_speed, min_space, time_pref, length) of a MicroVehicle; None = default_micro_vehicle(speed_limit)."""
    if v is None:
        from road.vehicle.micro_vehicle import MicroVehicle
        v = MicroVehicle.default_micro_vehicle(sim.speed_limit)
    return [float(v.accel_max), float(v.accel_pref), float(v.target_speed), float(v.min_space), float(v.time_pref), float(v.length)]


class ItscpEnv:

    def __init__(self, schedule_callback=itscp_random_schedule):
        self.schedule_callback = schedule_callback
        self.config = dict(default_config)
        self.simulator = None
        self.lane = {}
        self.schedule = {}
        self.macro_route_schedule = []
        self.queue_length = {}
        self.flux = {}
        self.avg_speed = []
        self.is_static_rms = RunningMean(100_000)
        self.render_eval = False
        self.route_provider = None          # optional callable(lane_id) -> MicroRoute replacing create_random_route
        self.time = self.steps = 0

    def action_size(self):
        length = self.config["policy_length"] * self.config["duration"]
        return int(length / self.config["signal_length"]) * (self.num_intersection ** 2)

    def reset(self):
        if self.config["random_seed"] > 0:
            np.random.seed(self.config["random_seed"])
        self.num_intersection = self.config["num_intersection"]
        self.num_lane = self.config["num_lane"]
        self.num_timestep = self.config["policy_length"] * self.config["duration"] * self.config["simulation_frequency"]
        self._make_road()
        self.schedule = self.schedule_callback(list(self.lane.keys()), self.num_timestep)
        self.observation_space = Box(0, 1, shape=(self.config["num_schedule_obs"] * len(self.lane),))
        self.action_space = Box(self.config["action_min"], self.config["action_max"], shape=(self.action_size(),))
        self.time = self.steps = 0
        self.reward_queue_c = -1.0
        self.macro_route_schedule = [self.simulator.create_random_macro_route() for _ in range(self.num_timestep)]
        self._make_micro_route()
        self._fused_cache = None            # tables of the fused kernels depend on the schedules / routes drawn above
        self._fused_done = False
        return self.observe()

    def rewind(self):
        """Episode state back to what reset() left, keeping the drawn schedules, routes and the uploaded kernel tables.
        Valid after fused episodes only (they never touch the lane objects); the reference deep-copies the environment
        per episode instead (trainer.py:172)."""
        if self.steps and not getattr(self, "_fused_done", False):
            raise RuntimeError("rewind() after a lane-by-lane episode: the lane objects moved, call reset()")
        self.time = self.steps = 0
        self._fused_done = False
        self.queue_length.clear()
        self.flux.clear()
        self.is_static_rms = RunningMean(100_000)

    def episode_copy(self):
        """A twin for ONE episode that must not disturb this environment (Trainer.evaluate; the reference deep-copies the
        environment, trainer.py:172): the episode state is its own, everything else is shared -- a fused episode never touches
        the lane objects.  Should the episode have to run lane by lane after all, the twin takes its own copy of the lanes first
        (step -> _own_lanes)."""
        twin = object.__new__(type(self))
        twin.__dict__.update(self.__dict__)
        twin.queue_length, twin.flux = {}, {}
        twin.config = dict(self.config)
        twin.is_static_rms = RunningMean(100_000)
        twin._lanes_shared = True
        return twin

    def _own_lanes(self):
        if getattr(self, "_lanes_shared", False):
            import copy
            memo = {}
            self.simulator = copy.deepcopy(self.simulator, memo)
            self.lane = copy.deepcopy(self.lane, memo)
            self._lanes_shared = False

    def __deepcopy__(self, memo):
        """Episode copy for the lane-by-lane path: everything is copied except the uploaded tables of the fused kernels,
        which are immutable and shared."""
        import copy
        twin = object.__new__(type(self))
        memo[id(self)] = twin
        # what an episode only reads is shared, not copied: the drawn inflow schedules (lanes x steps numpy scalars -- 86 000 objects
        # at config 4, 0.4 s per copy) and the per-step macro routes, beside the uploaded tables
        shared = ("_fused_cache", "_batched_net", "schedule", "macro_route_schedule")
        for k, v in self.__dict__.items():
            twin.__dict__[k] = v if k in shared else copy.deepcopy(v, memo)
        return twin

    def _make_micro_route(self):
        sim = self.simulator
        sim.lane_waiting_micro_vehicle.clear()
        sim.lane_waiting_micro_route.clear()
        for lid in sim.lane.keys():
            pairs = [sim.create_default_vehicle_with_random_route(lid) for _ in range(self.config["max_num_micro_vehicle_per_lane"])]
            sim.lane_waiting_micro_vehicle[lid] = [p[0] for p in pairs]
            sim.lane_waiting_micro_route[lid] = [p[1] for p in pairs]

    # ---- topology --------------------------------------------------------------------------------------------------
    def _make_road(self):
        n_int, n_lane = self.config["num_intersection"], self.config["num_lane"]
        outer = (LANE_WIDTH + 10) + LANE_WIDTH * (n_lane - 3 + 0.5)
        access = self.config["lane_length"]
        self.simulator = SimRoadNetwork(self.config["speed_limit"])
        if self.route_provider is not None:
            self.simulator.create_random_route = self.route_provider
        self.lane.clear()
        for row in range(n_int):
            for col in range(n_int):
                center = np.array([col * (outer + access), row * (outer + access)]) * 2.0
                approaching_ids = []
                for corner in range(4):
                    ang = np.radians(90 * corner)
                    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
                    for approaching in (True, False):
                        loc = SIDE_OF_CORNER[corner][0 if approaching else 1]
                        for k in range(n_lane):
                            lid = LaneID(row, col, loc, None, approaching, k)
                            a = np.array([LANE_WIDTH * (k + 0.5), access + outer])
                            b = np.array([LANE_WIDTH * (k + 0.5), outer])
                            if not approaching:
                                a, b = np.flip(a, axis=0), np.flip(b, axis=0)
                            self._make_lane(lid, center + rot @ a, center + rot @ b)
                            if approaching:
                                approaching_ids.append(lid)
                idx = 0
                for lid in approaching_ids:
                    seg = self.lane[lid].env_lane
                    for turn in ("straight", "right"):
                        if turn == "right" and lid.lane_id != n_lane - 1:
                            continue
                        n_loc = (RIGHT if turn == "right" else STRAIGHT)[lid.loc]
                        nid = LaneID(row, col, n_loc, None, False, lid.lane_id)
                        nseg = self.lane[nid].env_lane
                        mid = LaneID(row, col, "mid", lid.loc, True, idx)
                        idx += 1
                        self._make_lane(mid, seg.position(seg.length), nseg.position(nseg.length))
                        self._connect(lid, mid)
                        self._connect(mid, nid)
        for row in range(n_int):
            for col in range(n_int):
                for side, other, dr, dc in (("north", "south", -1, 0), ("west", "east", 0, -1)):
                    if (side == "north" and row == 0) or (side == "west" and col == 0):
                        continue
                    for approaching in (True, False):
                        for k in range(n_lane):
                            cur = LaneID(row, col, side, None, approaching, k)
                            con = LaneID(row + dr, col + dc, other, None, not approaching, k)
                            if approaching:
                                self._connect(con, cur)
                            else:
                                self._connect(cur, con)

    def _make_lane(self, lid, start, end):
        seg = _Segment(start, end)
        n_int, sl, dx, mode = self.config["num_intersection"], self.config["speed_limit"], self.config["cell_length"], self.config["mode"]
        sid = len(self.simulator.lane)
        if mode == "macro":
            sim_lane = dMacroLane(sid, seg.length, sl, dx)
        elif mode == "micro":
            sim_lane = MicroLane(sid, seg.length, sl)
        else:       # hybrid: intersections on the border of the grid are macro, interior ones micro
            border = lid.row in (0, n_int - 1) or lid.col in (0, n_int - 1)
            sim_lane = dMacroLane(sid, seg.length, sl, dx) if border else dMicroLane(sid, seg.length, sl)
        self.simulator.add_lane(sim_lane)
        self.lane[lid] = _Lane(seg, sim_lane)

    def _connect(self, a, b):
        self.simulator.connect_lane(self.lane[a].sim_lane.id, self.lane[b].sim_lane.id)

    # ---- observation / step ------------------------------------------------------------------------------------------
    def observe(self):
        n_obs = self.config["num_schedule_obs"]
        obs = []
        for lid in self.lane.keys():
            sc = self.schedule[lid]
            t = len(sc) // n_obs
            for k in range(n_obs):
                if len(self.lane[lid].sim_lane.prev_lane) == 0:
                    t0, t1 = int(t * k), min(int(t * k + t), len(sc))
                    obs.append(sum(sc[t0:t1]) / (t1 - t0))
                else:
                    obs.append(0)
        return np.array(obs).astype(np.float32)

    def step(self, action, differentiable):
        self.steps += 1
        self.queue_length.clear()
        self.flux.clear()
        if getattr(self, "_fused_done", False):
            raise NotImplementedError("the fused episode leaves the lane objects at their reset state: call reset() first, "
                                      "or set config['fused'] = False to step lane by lane")
        reward = self._step_fused(action, differentiable)
        if reward is None:
            self.last_path = "lane-by-lane"
            self._own_lanes()                       # (an episode_copy() twin: the lane-by-lane path moves the lane objects)
            self._simulate(action, differentiable)
            reward = self._reward(action)
        obs = self.observe()
        info = {"img": []}
        return obs, reward, self.steps >= self.config["duration"], info

    # ---- fused episode: the whole differentiable rollout in two kernel launches (dhts_net_*_rollout_fwd / _bwd) -----------
    def _step_fused(self, action, differentiable=True):
        """First step after reset() in `macro` / `hybrid` mode with config["fused"] (default on): reward (differentiable
        w.r.t. `action`) and the per-step queue terms from the fused network kernels instead of one operator call per
        lane and step.  differentiable=False (an evaluation episode, Trainer.evaluate): the same episode with the reference's
        hard thresholds (dhts_net_*_rollout_eval), one launch, nothing kept for a reverse sweep.  Same numbers as the operator path in `macro` mode; in `hybrid` mode vehicle routes are pre-drawn
        per spawn lane (`fused_routes`, or 8 per lane from create_random_route) instead of being drawn at spawn time.
        Returns None when the network or the call is outside what the kernels cover (the operator path runs then)."""
        if not self.config.get("fused", True) or self.config["mode"] not in ("macro", "hybrid", "micro") or self.steps != 1 or self.time != 0:
            return None
        if not (isinstance(action, th.Tensor) and action.is_cuda):
            return None
        from dhts import _lib, ops
        sim = self.simulator
        if any(sl.is_micro() and sl.num_vehicle() for sl in sim.lane.values()):
            return None
        cache = getattr(self, "_fused_cache", None)
        lane_cap = getattr(self, "_fused_lane_capacity", 0)     # vehicles a micro lane holds (0 = the kernels' default 16)
        if cache is None:
            cache = self._fused_cache = self._build_fused_cache(action.device, lane_cap)
        kind, tab = cache
        if kind == "none":
            return None
        args = (self.num_intersection ** 2, self.config["signal_length"] * self.config["simulation_frequency"],
                1.0 / self.config["simulation_frequency"], self.simulator.speed_limit, self.config["static_speed"],
                self.simulator.vehicle_length)
        a = action.reshape(1, -1)
        draws = None
        if self.config["mode"] == "micro" and kind in ("micro", "stepwise"):
            draws = getattr(self, "fused_draws", None)
            pending = self.__dict__.pop("_fused_pending_draws", None)
            if pending is not None:
                draws = pending
            elif draws is None:
                draws = np.random.random(self._fused_n_draws)
            else:
                draws = np.concatenate([np.asarray(draws, dtype=np.float64), np.full(self._fused_n_draws, 2.0)])[:self._fused_n_draws]
            tab.set_draws(draws)
        try:
            if kind == "batched":
                if self.config.get("batched_graph", True) and not getattr(self, "_batched_graph_failed", False):
                    try:                                    # the whole episode as one HIP graph (captured at the first call)
                        reward, queue = tab.graphed_rollout(a[0], *args, differentiable=differentiable)
                    except RuntimeError as e:
                        if isinstance(e, ops.CapacityError) or "capture" not in str(e).lower():
                            raise
                        import warnings
                        warnings.warn("ItscpEnv: HIP-graph capture of the batched episode failed (%s); running it eagerly" % e)
                        self._batched_graph_failed = True
                        reward, queue = tab.rollout(a[0], *args, differentiable=differentiable)
                else:
                    reward, queue = tab.rollout(a[0], *args, differentiable=differentiable)
                reward, queue = reward.reshape(1), queue.unsqueeze(0)
            elif kind == "stepwise":
                # a network beyond one workgroup (or an episode beyond the fused kernels' vehicle capacities): step by step on the
                # device, all lanes at once (dhts/stepwise.py)
                cut, _, queue, counts = tab.rollout(a[0], *args, differentiable=differentiable)
                reward, queue = cut.reshape(1), queue.unsqueeze(0)
                self.fused_counts = counts.tolist()
            elif kind == "macro":
                reward, queue = ops.net_macro_rollout(a, tab, *args) if differentiable else ops.net_macro_eval(a, tab, *args)
            elif differentiable:
                reward, _, queue, counts = ops.net_hybrid_rollout(a, tab, *args)
                self.fused_counts = counts[0].tolist()
            else:
                reward, queue, counts = ops.net_hybrid_eval(a, tab, *args)
                self.fused_counts = counts[0].tolist()
        except _lib.DhtsError as e:
            # Two library errors have another way to run, both DHTS_E_INVALID sizing refusals: the persistent form of a network whose
            # scratch does not fit a workgroup's LDS (-> the stepwise form), and a fused hybrid launch whose LDS plan does not fit at
            # the lane capacity the ladder asked for (-> the next rung, below).  Anything else -- a failed launch, a bad argument, any
            # error of the macro paths -- is a bug or a broken device and goes to the caller.
            if e.status != _lib.E_INVALID or kind not in ("hybrid", "micro", "stepwise"):
                raise
            if kind == "stepwise":
                if not getattr(tab, "persistent", False):
                    raise
                self._stepwise_no_persistent = True
                self._stepwise_cache = None
                self._fused_cache = None
                if draws is not None:
                    self._fused_pending_draws = draws
                return self._step_fused(action, differentiable)
            return self._climb_capacity_ladder(action, differentiable, kind, lane_cap, draws, e)
        except ops.CapacityError as e:
            if kind == "stepwise" and e.index == -2:
                # the hand-off event list (dhts_netstep_tables::max_events), not a lane: more lanes' worth of vehicles would not help.
                # The hard bound: every micro lane's head leaves (with up to three deposit cells) and every capacitor spawns, every step
                bound = self.num_timestep * (4 * tab.n_micro + 2 * tab.n_caps) + 64
                have = tab.max_events if tab.max_events > 0 else 0
                if have < bound:
                    self._stepwise_max_events = bound
                    self._stepwise_cache = None
                    self._fused_cache = None
                    if draws is not None:
                        self._fused_pending_draws = draws
                    return self._step_fused(action, differentiable)
            return self._climb_capacity_ladder(action, differentiable, kind, lane_cap, draws, e)
        self.last_path = {"macro": "fused", "hybrid": "fused", "micro": "fused"}.get(kind, kind)
        q = np.ascontiguousarray(queue[0].detach().cpu().numpy().T)      # [L][T]
        for i, lid in enumerate(self.lane.keys()):
            self.queue_length[lid] = q[i].tolist()              # (Python floats like the lane-by-lane path's, converted in C)
            self.flux.setdefault(lid, [])
        self.time = self.num_timestep
        self._fused_done = True
        return (-self.reward_queue_c) * reward[0]

    def _climb_capacity_ladder(self, action, differentiable, kind, lane_cap, draws, e):
        """The same episode on the next rung (more vehicle slots per micro lane, then the stepwise path), or lane by lane (None)."""
        # The episode needs more than this launch was sized for (vehicles per micro lane, vehicles per episode, records), or the
        # sizing does not fit one workgroup's LDS (DhtsError).  Nothing on the host was touched by the attempt; the ladder is
        #   fused kernels at 16 vehicles per lane -> fused at 128 -> stepwise device path at 32 -> 128 -> 1024 -> lane by lane
        #   (a network that starts on the stepwise path starts at the capacity its geometry asks for, dhts.stepwise.default_lane_capacity),
        # every rung the SAME episode: same drawn routes (kept in _fused_routes_drawn), same admission draws.
        ladder = [("fused", 16), ("fused", 128), ("stepwise", 32), ("stepwise", 128), ("stepwise", 1024)]
        here = ("stepwise" if kind == "stepwise" else "fused",
                lane_cap if lane_cap else (16 if kind != "stepwise" else getattr(self, "_stepwise_lane_capacity", 32)))
        nxt = None
        if kind in ("micro", "hybrid", "stepwise"):
            for rung in ladder:
                if (rung[0] == "stepwise", rung[1]) > (here[0] == "stepwise", here[1]) and rung[1] <= int(self.config.get("fused_max_lane_capacity", 1024)):
                    nxt = rung
                    break
        if nxt is not None:
            self._fused_lane_capacity = nxt[1]
            self._fused_prefer_stepwise = nxt[0] == "stepwise"
            self._fused_cache = None
            if draws is not None:
                self._fused_pending_draws = draws          # the retry is the same episode: the same admission draws
            return self._step_fused(action, differentiable)
        # The reference has no such limits (_micro_lane.py:53-113): this episode runs lane by lane instead (minutes, not
        # milliseconds).  In `micro` mode the admission draws the kernels were given are replayed, so that the episode is the one
        # that was asked for.
        if not getattr(self, "_fused_overflow_warned", False):
            self._fused_overflow_warned = True
            import warnings
            warnings.warn("ItscpEnv: the device paths' capacity was exceeded (%s); this episode runs lane by lane" % e)
        self._own_lanes()                       # (an episode_copy() twin: from here on the lane objects are written to)
        if draws is not None:
            it = iter(np.asarray(draws, dtype=np.float64).tolist())

            def replay():
                v = next(it, None)
                return float(np.random.random()) if v is None else v
            self.simulator.random_draw = replay
        self.fused_overflowed = True
        return None

    def _build_fused_cache(self, device, lane_cap):
        """(kind, device tables) of this episode: "macro" / "hybrid" / "micro" = the fused kernels (one workgroup per network),
        "batched" / "stepwise" = the step-by-step device paths for networks beyond that (dhts/batched.py, dhts/stepwise.py),
        ("none", None) = lane by lane."""
        from dhts import ops
        from dhts.network import HybridNetworkTables, MacroNetworkTables
        sim = self.simulator
        mode = self.config["mode"]
        T = self.num_timestep
        try:
            n_cells = sum(getattr(sl, "num_cell", 0) for sl in sim.lane.values() if sl.is_macro())
            if mode == "macro" and n_cells + len(sim.lane) <= 1024:
                return ("macro", ops.DeviceNetTables(MacroNetworkTables.from_env(self), device))
            if mode == "macro":
                # more cells + lanes than one workgroup holds one item per thread of (e.g. --n_intersection=3 --n_lane=3: 360 lanes,
                # ~1 700 cells): step by step on the device (dhts/stepwise.py, persistent form: 6.9 ms per 120-step differentiable
                # episode of that network against 11.4 ms for the replayed HIP graph of round 4's batched-lane path, dhts/batched.py,
                # which config "macro_path" = "batched" still selects)
                tabs = MacroNetworkTables.from_env(self)
                if self.config.get("macro_path", "stepwise") == "stepwise":
                    return ("stepwise", self._stepwise_net(tabs, np.asarray([[-1, -1]], dtype=np.int32), device, 32))
                from dhts.batched import BatchedMacroNetwork
                net = getattr(self, "_batched_net", None)           # survives reset(): same topology, new schedules / routes
                try:
                    if net is None:
                        raise ValueError
                    net.update(tabs)                                # in place: an episode captured as a HIP graph stays valid
                except ValueError:
                    net = self._batched_net = BatchedMacroNetwork(tabs, device)
                return ("batched", net)
            tab, routes, veh_params = self._fused_episode_inputs()
            fits = True
            try:
                tab.check_kernel_limits()
            except ValueError:
                fits = False
            if fits and not getattr(self, "_fused_prefer_stepwise", False) and lane_cap in (0, 16, 32, 64, 128):
                return (mode, ops.DeviceHybridTables(tab, routes, device, lane_capacity=lane_cap, vehicle_params=veh_params))
            # beyond one workgroup (cells + lanes > 960, > 64 IDM lanes, > 16 spawning lanes) or beyond the fused kernels' vehicle
            # capacities: step by step on the device (dhts/stepwise.py)
            if not lane_cap:
                from dhts.stepwise import default_lane_capacity
                lane_cap = self._stepwise_lane_capacity = default_lane_capacity(tab, self.simulator.vehicle_length)
            return ("stepwise", self._stepwise_net(tab, routes, device, lane_cap, veh_params))
        except ValueError:
            return ("none", None)

    def _fused_episode_inputs(self):
        """(tables, route rows, per-row vehicle attributes or None) of a `hybrid` / `micro` mode episode as the network kernels take
        them -- host arrays; tools/probes/fuzz_env.py hands the same three to the CPU checker."""
        from dhts.network import HybridNetworkTables
        sim = self.simulator
        mode = self.config["mode"]
        T = self.num_timestep
        tab = HybridNetworkTables.from_env(self)
        veh_params = getattr(self, "fused_vehicle_params", None)     # [routes][6] beside `fused_routes` (hybrid mode), or None
        if mode == "micro":
            # every lane an IDM lane; source lanes admit their waiting vehicles against np.random draws
            # (_simulator.py:153-174): the waiting routes in admission order (the list is popped from its end) are
            # the route rows, the draws of the episode are drawn up front (fused_draws replays a recorded stream)
            rows, vrows = [], []
            for l in range(tab.n_lanes):
                waiting = sim.lane_waiting_micro_vehicle.get(l, [])
                for k, r in enumerate(reversed(sim.lane_waiting_micro_route.get(l, []))):
                    r = list(r.route)[:32]
                    rows.append(r + [-1] * (32 - len(r)))
                    v = waiting[len(waiting) - 1 - k] if k < len(waiting) else None
                    vrows.append(_vehicle_attributes(v, sim))
            routes = np.asarray(rows if rows else [[-1, -1]], dtype=np.int32)
            # the waiting vehicles' own IDM attributes ride beside their routes (dhts_hybrid_tables::veh_params) unless every one of
            # them is the default vehicle the reference's reset() builds (_env.py:205-219)
            if any(v != _vehicle_attributes(None, sim) for v in vrows):
                veh_params = np.asarray(vrows, dtype=np.float64)
            self._fused_n_draws = T * max(1, int(tab.lane_source.sum()))
            tab.set_micro_sources(np.full(self._fused_n_draws, 2.0))
        else:
            routes = getattr(self, "fused_routes", None)
            if routes is None:
                routes = getattr(self, "_fused_routes_drawn", None)      # (a capacity retry is the same episode: the same routes)
            if routes is None:
                routes = []
                for l in range(tab.n_lanes):
                    if tab.lane_macro[l] == 0 and any(tab.lane_macro[a] for a in tab.prev_lanes[l]):
                        for _ in range(8):
                            r = list(sim.create_random_route(l).route)[:32]
                            routes.append(r + [-1] * (32 - len(r)))
                if not routes:
                    routes = [[-1, -1]]
                self._fused_routes_drawn = routes
            routes = np.asarray(routes, dtype=np.int32)
        return tab, routes, veh_params

    def _stepwise_net(self, tab, routes, device, lane_cap, veh_params=None):
        from dhts.stepwise import StepwiseNetwork
        net = getattr(self, "_stepwise_cache", None)            # survives reset(): same topology -> new per-episode tables only
        max_events = int(getattr(self, "_stepwise_max_events", 0))
        vkey = None if veh_params is None else np.asarray(veh_params, dtype=np.float64).tobytes()
        if net is not None and net[1] == (lane_cap, routes.shape, routes.tobytes(), vkey) and net[0].max_events == max_events:
            try:
                net[0].update(tab)
                if tab.lane_source.any():
                    net[0].t.draws = tab.draws
                return net[0]
            except ValueError:
                pass
        # persistent form (one kernel per direction) where it pays (up to 1 024 lanes / 4 096 cells: dhts.stepwise.persistent_form_pays)
        # unless config["stepwise_persistent"] says otherwise; a network whose scratch does not fit one workgroup's LDS comes back as
        # DhtsError at the first launch and is rebuilt in the stepwise form (a handful of launches per step)
        from dhts.stepwise import persistent_form_pays
        persistent = bool(self.config.get("stepwise_persistent", persistent_form_pays(tab))) and not getattr(self, "_stepwise_no_persistent", False)
        sw = StepwiseNetwork(tab, routes, device, lane_capacity=lane_cap, persistent=persistent, max_events=max_events, vehicle_params=veh_params)
        self._stepwise_cache = (sw, (lane_cap, routes.shape, routes.tobytes(), vkey))
        return sw

    def _simulate(self, action, differentiable):
        self.time = 0
        for _ in range(self.num_timestep):
            self._simulate_step(action, differentiable)
        return []

    def _is_static(self, speed, differentiable):
        """sigmoid(k (static_speed - speed)), k = 16 / |running mean of all (static_speed - speed) seen so far|, one
        sample at a time in visiting order (reference _env.py:586-618); `speed` is a 1-D tensor of one lane."""
        s0 = self.config["static_speed"]
        if not differentiable:
            return (speed < s0).float()
        with th.no_grad():
            means = self.is_static_rms.prefix_means((s0 - speed).detach().cpu().numpy())
            k = th.as_tensor(16.0 / np.abs(means), dtype=th.float32, device=speed.device)
        return th.sigmoid(th.clamp((s0 - speed) * k, -16.0, 16.0))

    def _simulate_step(self, action, differentiable):
        frame = self.time
        sim = self.simulator
        for lid in self.lane.keys():
            sl = self.lane[lid].sim_lane
            sim.lane_incoming[sl.id] = self.schedule[lid][frame] if len(sl.prev_lane) == 0 else -1
            sim.lane_signal[sl.id] = self.lane_signal_info(lid, action, frame, differentiable)[1]
        sim.macro_route = self.macro_route_schedule[frame]
        dt = 1.0 / self.config["simulation_frequency"]
        sim.forward(dt, differentiable)
        self.time += 1
        for lid in self.lane.keys():
            sl = self.lane[lid].sim_lane
            q = 0
            if isinstance(sl, MacroLane):
                r, _, u = sl.get_state_vector()
                q = (self._is_static(u, differentiable) * (r * sl.cell_length / sim.vehicle_length)).sum()
            elif isinstance(sl, MicroLane):
                if sl.num_vehicle():
                    _, v = sl.get_state_vector()
                    q = self._is_static(v, differentiable).sum()
            else:
                raise ValueError()
            self.queue_length.setdefault(lid, []).append((q ** 2.0) * dt)
            self.flux.setdefault(lid, [])

    def _reward(self, action):
        reward = 0
        for lid in self.lane.keys():
            for x in self.queue_length[lid]:
                reward = reward + self.reward_queue_c * x
        return reward

    # ---- signals -----------------------------------------------------------------------------------------------------
    def lane_signal_info(self, lane_id, action, curr_frame, differentiable):
        """(prev_signal, next_signal) of a lane (reference _env.py:885-962): within each signal phase the action value a
        of the lane's intersection splits the phase: west-east green while progress < a, north-south green afterwards."""
        frames = self.config["simulation_frequency"] * self.config["signal_length"]
        sq = self.num_intersection ** 2
        phase = min(curr_frame // frames, len(action) // sq - 1)
        a = action[phase * sq + lane_id.row * self.num_intersection + lane_id.col]
        if not isinstance(a, th.Tensor):
            a = th.tensor(a)
        progress = min((curr_frame % frames) / frames, 1.0)

        def we():
            return sigmoid(a - progress, constant=32) if differentiable else float(a > progress)

        def ns():
            return sigmoid(progress - a, constant=32) if differentiable else float(progress > a)

        if lane_id.loc == "mid":
            return (we() if lane_id.ploc in ("west", "east") else ns()), 1.0
        if not lane_id.approaching:
            return 1.0, 1.0
        return 1.0, (we() if lane_id.loc in ("west", "east") else ns())
