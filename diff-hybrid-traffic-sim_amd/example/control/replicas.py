"""R itscp environments of ONE topology as one batch (this build only; the reference trains on one environment per episode,
example/control/trainer.py:168-205).

The fused network kernels run one workgroup per replica (dhts_net_macro_rollout_* / dhts_net_hybrid_rollout_*; BASELINE configs
4-5: 256 replicas per GPU), and a launch sized for 256 replicas takes as long as one for a single replica -- a trainer that runs
ONE environment per episode leaves 255 of 256 compute units idle and pays its host time per episode.  `ReplicaBatch` holds R
environments that share the topology and differ in what reset() draws (inflow schedules, per-step macro routes, admission draws):
their tables go to the device once as per-replica tables (dhts.ops.DeviceNetTables / DeviceHybridTables take a list), and an
"episode" of the batch is [R][A] actions -> [R] rewards in two launches.  Networks the fused kernels cannot hold run as R workgroups
of the stepwise path's persistent kernels (dhts/stepwise.py: StepwiseNetwork over a list of tables), same interface; only what
neither holds falls back to one ItscpEnv.step per replica.
"""
import copy

import numpy as np
import torch as th


class ReplicaBatch:

    def __init__(self, env, n_replica, device, seed_stride=1, seed_offset=0):
        """env: an ItscpEnv after reset().  Replica 0 is `env` itself; replica r > 0 is a copy reset with random_seed =
        env's seed + seed_offset + r * seed_stride when the environment is seeded (> 0), else with whatever np.random yields next."""
        self.device = th.device(device)
        self.R = int(n_replica)
        base = int(env.config.get("random_seed", 0))
        self.envs = [env]
        for r in range(1, self.R):
            e = copy.deepcopy(env)
            e.__dict__.pop("_fused_cache", None)
            if base > 0:
                e.config["random_seed"] = base + seed_offset + r * seed_stride      # (the copy's own dict)
            e.reset()
            self.envs.append(e)
        if base > 0 and seed_offset:
            # replica 0 of another rank draws its own episode too -- with the caller's configuration left as it was (a second
            # batch built from the same environment must not see an accumulated offset)
            env.config["random_seed"] = base + seed_offset
            try:
                env.reset()
            finally:
                env.config["random_seed"] = base
        self.kind, self.tab = self._build()
        e0 = self.envs[0]
        self.args = (e0.num_intersection ** 2, e0.config["signal_length"] * e0.config["simulation_frequency"],
                     1.0 / e0.config["simulation_frequency"], e0.simulator.speed_limit, e0.config["static_speed"], e0.simulator.vehicle_length)

    def _build(self):
        from dhts import ops
        from dhts.network import HybridNetworkTables, MacroNetworkTables
        e0 = self.envs[0]
        mode = e0.config["mode"]
        try:
            from dhts.stepwise import StepwiseNetwork, default_lane_capacity
            if mode == "macro":
                tabs = [MacroNetworkTables.from_env(e) for e in self.envs]
                if tabs[0].n_cells + tabs[0].n_lanes > 1024:
                    return "stepwise", StepwiseNetwork(tabs, np.asarray([[-1, -1]], dtype=np.int32), self.device, persistent=True)
                return "macro", ops.DeviceNetTables(tabs if self.R > 1 else tabs[0], self.device)
            tabs = [HybridNetworkTables.from_env(e) for e in self.envs]
            sim = e0.simulator
            if mode == "micro":
                # ONE route table for the batch (the kernels' routes are shared by the replicas): replica 0's waiting lists.  Every
                # replica environment is given the same lists, so that the per-environment path of a comparison -- or of the
                # fall-back -- runs the episode the batch runs; what differs between replicas is what the kernels take per replica:
                # inflow schedules and admission draws.
                rows = []
                for l in range(tabs[0].n_lanes):
                    for r in reversed(sim.lane_waiting_micro_route.get(l, [])):
                        r = list(r.route)[:32]
                        rows.append(r + [-1] * (32 - len(r)))
                routes = np.asarray(rows if rows else [[-1, -1]], dtype=np.int32)
                for e in self.envs[1:]:
                    e.simulator.lane_waiting_micro_route = copy.deepcopy(sim.lane_waiting_micro_route)
                self.n_draws = e0.num_timestep * max(1, int(tabs[0].lane_source.sum()))
                for t in tabs:
                    t.set_micro_sources(np.full(self.n_draws, 2.0))       # (placeholders: rollout() draws per episode)
            else:
                routes = getattr(e0, "fused_routes", None)
                if routes is None:
                    routes = []
                    for l in range(tabs[0].n_lanes):
                        if tabs[0].lane_macro[l] == 0 and any(tabs[0].lane_macro[a] for a in tabs[0].prev_lanes[l]):
                            for _ in range(8):
                                r = list(sim.create_random_route(l).route)[:32]
                                routes.append(r + [-1] * (32 - len(r)))
                    routes = routes or [[-1, -1]]
                routes = np.asarray(routes, dtype=np.int32)
                for e in self.envs:
                    e.fused_routes = routes              # (the per-environment path of a comparison sees the same routes)
            try:
                tabs[0].check_kernel_limits()
            except ValueError:
                return "stepwise", StepwiseNetwork(tabs, routes, self.device, lane_capacity=int(e0.config.get("stepwise_lane_capacity", 0)) or default_lane_capacity(tabs[0], e0.simulator.vehicle_length),
                                                   persistent=True)
            return mode, ops.DeviceHybridTables(tabs if self.R > 1 else tabs[0], routes, self.device)
        except ValueError:
            return "per-env", None

    def observe(self):
        """[R][n_obs].  ItscpEnv.observe() is a function of the drawn inflow schedules alone (reference _env.py:541-558), i.e. constant
        until the next reset(): evaluated once per replica (144 lanes x 5 windows of Python per call otherwise: 0.5 ms per replica)."""
        if getattr(self, "_obs", None) is None:
            self._obs = np.stack([e.observe() for e in self.envs])
        return self._obs

    def rollout(self, actions, differentiable=True):
        """actions [R][A] (device) -> rewards [R] (reward_queue_c applied like ItscpEnv._reward); differentiable w.r.t. actions."""
        from dhts import ops
        c = -self.envs[0].reward_queue_c
        if self.kind != "per-env" and getattr(self, "n_draws", 0):
            # itscp `micro` mode: fresh admission draws for every episode and replica, as ItscpEnv._step_fused draws them per episode
            # (`fused_draws` [R][n] replays a recorded stream)
            d = getattr(self, "fused_draws", None)
            self.last_draws = np.random.random((self.R, self.n_draws)) if d is None else np.asarray(d, dtype=np.float64)
            self.tab.set_draws(self.last_draws if self.R > 1 or self.kind == "stepwise" else self.last_draws[0])
        if self.kind == "per-env":
            out = []
            for r, e in enumerate(self.envs):
                e.rewind() if getattr(e, "_fused_done", False) else None
                _, reward, _, _ = e.step(actions[r], differentiable)
                out.append(reward.reshape(()) if isinstance(reward, th.Tensor) else th.as_tensor(float(reward), device=self.device))
            return th.stack(out)
        if self.kind == "stepwise":
            reward, _, _, _ = self.tab.rollout(actions, *self.args, differentiable=differentiable)
            return c * reward
        if self.kind == "macro":
            reward, _ = ops.net_macro_rollout(actions, self.tab, *self.args) if differentiable else ops.net_macro_eval(actions, self.tab, *self.args)
        elif differentiable:
            reward, _, _, _ = ops.net_hybrid_rollout(actions, self.tab, *self.args)
        else:
            reward, _, _ = ops.net_hybrid_eval(actions, self.tab, *self.args)
        return c * reward

    @property
    def path(self):
        return "per-env" if self.kind == "per-env" else ("stepwise x%d" % self.R if self.kind == "stepwise" else "fused x%d" % self.R)
