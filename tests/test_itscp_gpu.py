"""The itscp environment (differentiable traffic-signal control) through the mirror classes on the GPU, against the
reference's own runs (G8, tools/gen_goldens.py): lane table, schedules, reward, d reward / d action, per-step queues."""
import os

import numpy as np
import pytest

from util import TOL_GRAD, TOL_STATE, meta_of, rel_max, state_report

pytestmark = pytest.mark.gpu

# the lane-by-lane mirror path differs from the reference by the float32 glue torch evaluates on the GPU instead of the CPU
# (the operators themselves are the kernels); bounds = what the path achieves on each golden, rounded up
MIRROR_TOL = {"macro_small": TOL_GRAD, "macro": TOL_GRAD, "macro_2x2": TOL_GRAD, "hybrid_short": TOL_GRAD, "hybrid_p3": TOL_GRAD}


def build_env(g, m, replay_routes=True):
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp import problem as problems
    from road.network.route import MicroRoute
    env = ItscpEnv()
    env.schedule_callback = getattr(problems, "problem_%d" % int(m.get("problem", 1)))
    for k, v in dict(num_intersection=m["num_intersection"], lane_length=m["lane_length"], num_lane=m["num_lane"],
                     policy_length=m["policy_length"], signal_length=m["signal_length"], mode=m["mode"],
                     speed_limit=m["speed_limit"], random_seed=m["seed"]).items():
        env.config[k] = v
    spawn = [MicroRoute([int(x) for x in r if x >= 0]) for r in g["spawn_routes"]]
    cursor = {"i": 0}

    def provider(lane_id):
        # routes drawn at spawn time come from the recorded table; the ones drawn at reset are discarded by the reference
        # as well (waiting vehicles only exist for source micro lanes)
        if env.time == 0 and not getattr(env, "_armed", False):
            return MicroRoute([lane_id])
        r = spawn[cursor["i"]]
        cursor["i"] += 1
        return r
    if replay_routes:
        env.route_provider = provider
    env.reset()
    env._armed = True
    return env


@pytest.mark.parametrize("name", ["macro_small", "macro", "macro_2x2", "hybrid_short", "hybrid_p3", "hybrid"])
def test_itscp_rollout_matches_reference(cuda, golden_dir, name):
    import torch
    path = os.path.join(golden_dir, "itscp_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    if name == "hybrid" and not os.environ.get("DHTS_SLOW"):
        pytest.skip("144 lanes x 600 steps through per-lane operator calls takes minutes; set DHTS_SLOW=1")
    g = np.load(path)
    m = meta_of(g)
    env = build_env(g, m)
    keys = list(env.lane.keys())
    tab = g["lane_tab"]
    # topology: same lanes in the same order with the same lengths, cell counts and connectivity
    assert len(keys) == len(tab) and env.num_timestep == m["T"]
    for i, k in enumerate(keys):
        sl = env.lane[k].sim_lane
        assert sl.id == i and float(sl.is_macro()) == tab[i, 1]
        assert abs(sl.length - tab[i, 2]) <= 1e-12 * max(1.0, tab[i, 2])
        assert getattr(sl, "num_cell", 0) == int(tab[i, 3])
        assert "%s|%s|%d" % (k.loc, k.ploc, int(k.approaching)) == str(g["lane_str"][i])
        assert (k.row, k.col, k.lane_id) == tuple(int(x) for x in tab[i, 5:8])
    edges = sorted((a, b) for a in env.simulator.lane for b in env.simulator.lane[a].next_lane.keys())
    assert edges == sorted(tuple(e) for e in g["edges"].tolist())
    # host RNG parity: the schedule and the per-step macro routes are drawn in the reference's order
    sched = np.array([env.schedule[k] for k in keys])
    assert np.array_equal(sched, g["schedule"])
    mr = -np.ones_like(g["macro_route"])
    for t, r in enumerate(env.macro_route_schedule):
        for a, b in r.next_lane_dict.items():
            mr[t, a] = b
    assert np.array_equal(mr, g["macro_route"])
    # rollout
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    env._simulate(action, True)
    reward = env._reward(action)
    queue = np.array([[float(x) for x in env.queue_length[k]] for k in keys])
    assert env.simulator.num_vehicle == m["n_vehicle_spawned"]
    assert state_report("mirror path %s: queues vs reference" % name, queue, g["queue"]) <= TOL_STATE
    assert abs(float(reward) - float(g["reward"])) <= TOL_STATE * abs(float(g["reward"]))
    if name != "hybrid":
        reward.backward()
        e = rel_max(action.grad.cpu().numpy(), g["g_action"])
        print("mirror path %s: d reward / d action error / max|g| = %.2e" % (name, e))
        assert e <= MIRROR_TOL[name]
        return
    # ---- the 600-step hybrid case -------------------------------------------------------------------------------
    # From step 480 lane 16 (one cell) holds a deposited standing vehicle behind a red light; the reverse sweep amplifies the
    # cotangent of that lane's late loss terms by 1.144 per step over the ~90 standing steps before them, so the gradient of a
    # PART of that loss is only defined up to a float32 lattice (tests/test_oracle_golden.py::
    # test_restricted_gradient_lattice_of_the_standing_vehicle); the parts below are the ones the reference's fixture holds
    # up to step 540, and they -- like the whole gradient -- are reproduced within the contract's 1e-4:
    def grad_of(part):
        if not (isinstance(part, torch.Tensor) and part.requires_grad):
            return np.zeros(len(g["action"]), np.float32)
        out = torch.autograd.grad(part, action, retain_graph=True, allow_unused=True)[0]
        return np.zeros(len(g["action"]), np.float32) if out is None else out.cpu().numpy()

    def neg_sum(xs):
        tot = 0
        for x in xs:
            tot = tot + (-1.0) * x
        return tot
    T = m["T"]
    scale = np.abs(g["g_action"]).max()
    # (1) reward restricted to its first t0 steps, t0 <= 540: spawns (steps 59-400), lane changes, all 12 deposits
    for t0, ref in zip(g["g_action_cut_steps"], g["g_action_cut"]):
        if t0 <= 540:
            mine = grad_of(neg_sum(x for k in keys for x in env.queue_length[k][:int(t0)]))
            print("mirror hybrid: cut %d error / max|g| = %.2e" % (t0, np.abs(mine - ref).max() / scale))
            assert np.abs(mine - ref).max() <= TOL_GRAD * scale, int(t0)
    # (2) the last-quarter loss of every other lane that receives deposits
    late = {int(i): ref for i, ref in zip(g["g_lane_late_ids"], g["g_lane_late"])}
    for i, ref in late.items():
        if i != 16:
            mine = grad_of(neg_sum(env.queue_length[keys[i]][(3 * T) // 4:]))
            print("mirror hybrid: late lane %d error = %.2e (of %.2e)" % (i, np.abs(mine - ref).max(), max(np.abs(ref).max(), 1e-3 * scale)))
            assert np.abs(mine - ref).max() <= TOL_GRAD * max(np.abs(ref).max(), 1e-3 * scale), i
    # (3) the full-horizon gradient without lane 16's last-quarter loss
    total = grad_of(neg_sum(x for k in keys for x in env.queue_length[k]))
    l16 = grad_of(neg_sum(env.queue_length[keys[16]][(3 * T) // 4:]))
    print("mirror hybrid: full %.2e  full - lane16 late %.2e  lane16 late %.2e (rel to its own max %.2e)" % (
        np.abs(total - g["g_action"]).max() / scale, np.abs((total - l16) - (g["g_action"] - late[16])).max() / scale,
        np.abs(l16 - late[16]).max() / scale, np.abs(l16 - late[16]).max() / np.abs(late[16]).max()))
    assert np.abs((total - l16) - (g["g_action"] - late[16])).max() <= TOL_GRAD * scale
    # (4) lane 16's own last-quarter loss (achieved 3.5e-7 of its maximum) and the whole gradient (3.4e-6)
    assert np.abs(l16 - late[16]).max() <= TOL_GRAD * np.abs(late[16]).max()
    assert np.abs(total - g["g_action"]).max() <= TOL_GRAD * scale


@pytest.mark.parametrize("name", ["micro_small", "micro", "micro_jam_a"])
def test_itscp_micro_mode_matches_reference(cuda, golden_dir, name):
    """itscp `micro` mode (run_itscp_micro.sh: every lane an IDM lane, source lanes admit waiting vehicles stochastically,
    _simulator.py:153-174) through the mirror classes on the kernels, against the reference's run.  The reference uses the
    plain autodiff MicroLane there (_env.py:484-487); the mirror's lanes run the analytic operator, which agrees with it to
    ~1e-6.  Host randomness is replayed: the admission draws in call order, the waiting routes by seed."""
    import json
    import torch
    path = os.path.join(golden_dir, "itscp_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path)
    m = meta_of(g)
    env = build_env(g, m, replay_routes=False)      # routes are drawn with np.random in the reference's order (same seed)
    sim = env.simulator
    keys = list(env.lane.keys())
    assert len(keys) == len(g["lane_tab"]) and all(env.lane[k].sim_lane.is_micro() for k in keys)
    # host RNG parity at reset: inflow schedule and the routes of the waiting vehicles, drawn in the reference's order
    assert np.array_equal(np.array([env.schedule[k] for k in keys]), g["schedule"])
    want = {int(l): r for l, r in json.loads(str(g["waiting_routes"])).items()}
    mine = {int(l): [list(r.route) for r in rs] for l, rs in sim.lane_waiting_micro_route.items()}
    assert mine == want
    draws = iter(g["rand_draws"].tolist())
    sim.random_draw = lambda: next(draws)
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    env._simulate(action, True)
    reward = env._reward(action)
    assert next(draws, None) is None                       # every recorded draw was consumed: same admission tests
    assert sim.num_vehicle == m["n_vehicle_spawned"]
    queue = np.array([[float(x) for x in env.queue_length[k]] for k in keys])
    # (the reference steps `micro` mode lanes with the autodiff MicroLane in float32 tensor arithmetic; so does the mirror's plain MicroLane in
    #  a differentiable episode -- dhts_micro_step_fwd_tensor --: 2.5e-7 / 3.8e-7 measured, 1.25e-5 / 3.7e-6 with the float64 ladder)
    assert state_report("mirror path %s: queues vs reference" % name, queue, g["queue"]) <= TOL_STATE
    assert abs(float(reward) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    reward.backward()
    e = rel_max(action.grad.cpu().numpy(), g["g_action"])
    print("mirror path %s: d reward / d action error / max|g| = %.2e" % (name, e))
    assert e <= TOL_GRAD


@pytest.mark.parametrize("name", ["macro_small", "macro", "macro_2x2", "macro_half", "macro_long", "hybrid_short", "hybrid_p2", "hybrid_p3", "hybrid_l10",
                                  "hybrid_half", "hybrid_s2", "hybrid_s3", "hybrid_p2_600"])
def test_env_step_uses_fused_kernels(cuda, golden_dir, name):
    """ItscpEnv.step(action, True) -- the reference's entry point (trainer.py:172-190) -- through the fused network
    kernels: reward, its gradient and the per-step queue terms against the reference's run."""
    import torch
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    m = meta_of(g)
    env = build_env(g, m)
    if name.startswith("hybrid"):
        env.fused_routes = g["spawn_routes"]
    keys = list(env.lane.keys())
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    obs, reward, done, info = env.step(action, True)
    assert env._fused_done and obs.shape == env.observe().shape
    reward.backward()
    queue = np.array([env.queue_length[k] for k in keys])
    assert state_report("env.step %s: queues vs reference" % name, queue, g["queue"]) <= TOL_STATE
    assert abs(float(reward.detach()) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert np.abs(action.grad.cpu().numpy() - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()
    if name.startswith("hybrid"):
        assert env.fused_counts[0] == m["n_vehicle_spawned"]
    with pytest.raises(NotImplementedError):
        env.step(action, True)
    env._armed = False
    env.reset()
    env.config["fused"] = False           # the lane-by-lane path stays available
    assert env._step_fused(action) is None


@pytest.mark.parametrize("name", ["eval_macro", "eval_macro_2x2", "eval_hybrid_short", "eval_hybrid"])
def test_env_evaluation_step_uses_fused_kernels(cuda, golden_dir, name):
    """ItscpEnv.step(action, False) -- what the reference's Trainer.evaluate calls every num_eval_epoch epochs
    (trainer.py:73-75, 94-142) -- through the fused evaluation kernels (one launch) instead of lane by lane: queues and
    reward of the reference's own evaluation episode; then the same episode lane by lane on a copy of the short cases (the
    mirror path the fused one replaces) gives the same numbers."""
    import copy
    import time
    import torch
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    m = meta_of(g)
    env = build_env(g, m)
    if "hybrid" in name:
        env.fused_routes = g["spawn_routes"]
    keys = list(env.lane.keys())
    slow = copy.deepcopy(env)
    action = torch.tensor(g["action"], device=cuda)
    with torch.no_grad():
        t0 = time.time()
        obs, reward, done, info = env.step(action, False)
        float(reward)
        t_fused = time.time() - t0
    assert env._fused_done
    queue = np.array([env.queue_length[k] for k in keys])
    assert rel_max(queue, g["queue"]) <= TOL_STATE
    assert abs(float(reward) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    if "hybrid" in name:
        assert env.fused_counts[0] == m["n_vehicle_spawned"]
    if name in ("eval_macro_2x2",):            # the lane-by-lane mirror of the same episode (seconds for a macro network)
        slow.config["fused"] = False
        with torch.no_grad():
            t0 = time.time()
            _, reward_slow, _, _ = slow.step(action, False)
            t_slow = time.time() - t0
        assert abs(float(reward_slow) - float(reward)) <= 1e-5 * abs(float(reward))
        print("evaluation episode %s: fused %.1f ms (first call, with table upload), lane by lane %.1f s" % (name, 1e3 * t_fused, t_slow))


@pytest.mark.parametrize("name", ["micro_small", "micro", "micro_jam_a"])
def test_env_micro_mode_step_uses_fused_kernels(cuda, golden_dir, name):
    """ItscpEnv.step(action, True) in `micro` mode through the fused kernels (two launches instead of 40 lanes x 300 steps of
    operator calls), the reference's recorded admission draws replayed as data (env.fused_draws)."""
    import torch
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    m = meta_of(g)
    env = build_env(g, m, replay_routes=False)      # waiting routes drawn with np.random in the reference's order (same seed)
    keys = list(env.lane.keys())
    env.fused_draws = g["rand_draws"]
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    obs, reward, done, info = env.step(action, True)
    assert env._fused_done and env.fused_counts[0] == m["n_vehicle_spawned"]
    reward.backward()
    queue = np.array([env.queue_length[k] for k in keys])
    # (`micro` mode: the fused kernels follow the reference's float32 tensor ladder there, idm_step_f32 -- measured 1.7e-7 / 2.2e-6)
    assert state_report("env.step %s: queues vs reference" % name, queue, g["queue"]) <= TOL_STATE
    assert abs(float(reward.detach()) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert np.abs(action.grad.cpu().numpy() - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()


def _crowded_micro_env(max_lane_capacity):
    """`micro`-mode episode with 150 m approach lanes, a light that is red nine tenths of the time for two of them and an
    admission at every opportunity: more than 16 vehicles stand on those lanes."""
    from example.control.itscp._env import ItscpEnv
    env = ItscpEnv()
    env.schedule_callback = lambda keys, T: {k: [1.0] * T for k in keys}            # inflow 1 everywhere: every draw admits
    for k, v in dict(num_intersection=1, num_lane=1, lane_length=150.0, policy_length=16, signal_length=2, mode="micro",
                     speed_limit=60.0, max_num_micro_vehicle_per_lane=30, random_seed=3,
                     fused_max_lane_capacity=max_lane_capacity).items():
        env.config[k] = v
    env.reset()
    env.fused_draws = np.zeros(env.num_timestep * 8)
    return env


def test_env_step_falls_back_when_a_fused_capacity_is_exceeded(cuda):
    """The fused kernels hold 16 vehicles per IDM lane unless a launch is sized for more (include/dhts.h: lane_capacity); the
    reference has no such limit (_micro_lane.py:53-113).  With the retry switched off (fused_max_lane_capacity = 16) the fused
    attempt of a crowded episode reports DHTS_FAULT_CAPACITY, ItscpEnv.step warns once and runs the episode lane by lane with
    the same admission draws.  With it (the default) the episode is retried with room for 128 vehicles per lane and stays on the
    fused kernels: same reward, same gradient."""
    import warnings
    import torch
    env = _crowded_micro_env(16)
    T = env.num_timestep
    action = torch.full((env.action_size(),), 0.1, device=cuda, requires_grad=True)   # west-east green for a tenth of each phase
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        obs, reward, done, info = env.step(action, True)
    assert getattr(env, "_fused_cache", ("none",))[0] == "micro"                        # the fused path was set up and tried
    assert getattr(env, "fused_overflowed", False) and not getattr(env, "_fused_done", False)
    assert any("capacity" in str(x.message) for x in w)
    keys = list(env.lane.keys())
    assert all(len(env.queue_length[k]) == T for k in keys)                             # the whole episode ran, lane by lane
    most = max(env.lane[k].sim_lane.num_vehicle() for k in keys)
    assert most > 16, most                                                              # ... past what the default launch holds
    assert torch.isfinite(reward.detach()).all()
    reward.backward()
    assert bool(torch.isfinite(action.grad).all()) and float(action.grad.abs().max()) > 0.0
    # the same episode with the retry: one failed launch at 16 vehicles per lane, then the fused kernels sized for 128
    env2 = _crowded_micro_env(128)
    action2 = torch.full((env2.action_size(),), 0.1, device=cuda, requires_grad=True)
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        _, reward2, _, _ = env2.step(action2, True)
    assert env2._fused_done and env2._fused_lane_capacity == 128 and not getattr(env2, "fused_overflowed", False)
    assert not any("capacity" in str(x.message) for x in w2)
    reward2.backward()
    q1 = np.array([[float(x) for x in env.queue_length[k]] for k in keys])
    q2 = np.array([[float(x) for x in env2.queue_length[k]] for k in keys])
    # (`micro` mode: both paths step these lanes in the reference's float32 tensor arithmetic)
    assert state_report("crowded micro episode, fused at 128 per lane vs lane by lane: queues", q2, q1) <= TOL_STATE
    assert abs(float(reward2.detach()) - float(reward.detach())) <= 1e-5 * abs(float(reward.detach()))
    g1, g2 = action.grad.cpu().numpy(), action2.grad.cpu().numpy()
    assert np.abs(g2 - g1).max() <= TOL_GRAD * np.abs(g1).max()


def test_episode_copy_leaves_the_environment_alone(cuda, golden_dir):
    """Trainer.evaluate runs its episodes on ItscpEnv.episode_copy() twins (the reference deep-copies the environment,
    trainer.py:172): a fused episode on the twin shares the lane objects, a lane-by-lane one takes its own copy first; either way
    the environment can be rewound and gives the same episode afterwards."""
    import torch
    g = np.load(os.path.join(golden_dir, "itscp_macro_small.npz"))
    m = meta_of(g)
    env = build_env(g, m)
    action = torch.tensor(g["action"], device=cuda)
    with torch.no_grad():
        _, r0, _, _ = env.step(action, False)
        env.rewind()
        twin = env.episode_copy()
        _, r1, _, _ = twin.step(action, False)                 # fused: nothing copied
        assert twin._fused_done and twin.simulator is env.simulator and float(r1) == float(r0)
        slow = env.episode_copy()
        slow.config["fused"] = False
        assert env.config.get("fused", True)                  # (the twin's configuration is its own)
        _, r2, _, _ = slow.step(action, False)                 # lane by lane: on the twin's own lanes
        assert slow.simulator is not env.simulator and not getattr(slow, "_fused_done", False)
        assert abs(float(r2) - float(r0)) <= 1e-5 * abs(float(r0))
        assert not any(sl.is_macro() and float(sl.get_state_vector()[0].abs().max()) > 0 for sl in env.simulator.lane.values())   # still empty
        env.rewind()
        _, r3, _, _ = env.step(action, False)
        assert float(r3) == float(r0)


@pytest.mark.parametrize("path", ["fused", "stepwise"])
def test_env_steps_waiting_vehicles_with_their_own_attributes(cuda, golden_dir, path):
    """Round 6: an ItscpEnv in `micro` mode whose waiting vehicles are NOT the default vehicle (here: the attributes of the reference
    run `micro_rv`, seeded MicroVehicle.random_micro_vehicle draws) hands them to the device paths beside the routes
    (dhts_hybrid_tables::veh_params): ItscpEnv.step reproduces the reference's reward and d reward / d action."""
    import json
    import torch
    g = np.load(os.path.join(golden_dir, "itscp_micro_rv.npz"))
    m = meta_of(g)
    env = build_env(g, m, replay_routes=False)
    sim = env.simulator
    want = {int(l): p for l, p in json.loads(str(g["waiting_params"])).items()}
    for l, lst in sim.lane_waiting_micro_vehicle.items():
        assert len(lst) == len(want[int(l)])
        for v, p in zip(lst, want[int(l)]):
            v.accel_max, v.accel_pref, v.target_speed, v.min_space, v.time_pref, v.length = p
    env.fused_draws = g["rand_draws"]
    if path == "stepwise":
        env._fused_prefer_stepwise, env._fused_lane_capacity = True, 32
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    _, reward, _, _ = env.step(action, True)
    reward.backward()
    assert env.last_path == path and env.fused_counts[0] == m["n_vehicle_spawned"]
    assert abs(float(reward.detach()) - float(g["reward"])) <= TOL_STATE * abs(float(g["reward"]))
    assert np.abs(action.grad.cpu().numpy() - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()
