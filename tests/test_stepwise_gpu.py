"""The stepwise device path for road networks of any size and any mix of lanes (dhts.stepwise.StepwiseNetwork over
dhts_netstep_rollout_fwd / _bwd, csrc/netstep_hybrid.hip) against the reference's own itscp runs (G8 goldens: macro, hybrid and
micro mode, training and evaluation episodes, networks inside AND beyond the fused kernels' limits), against the CPU oracle on
other actions, and against the fused kernels where both run."""
import os
import time

import numpy as np
import pytest

from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables
from util import TOL_GRAD, TOL_STATE, rel_max, state_report

pytestmark = pytest.mark.gpu


def _args(m):
    return (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])


def _net(cuda, g, lane_capacity=32, persistent=False):
    from dhts.stepwise import StepwiseNetwork
    t, m = itscp_hybrid_tables(g)
    routes = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
    return StepwiseNetwork(t, routes, cuda, lane_capacity=lane_capacity, persistent=persistent), t, m


def _run(cuda, net, m, action, loss_steps=0, differentiable=True):
    import torch
    a = torch.tensor(np.asarray(action, dtype=np.float32), device=cuda, requires_grad=differentiable)
    cut, reward, queue, counts = net.rollout(a, *_args(m), differentiable=differentiable, loss_steps=loss_steps)
    grad = None
    if differentiable:
        cut.backward()
        grad = a.grad.cpu().numpy()
    return dict(cut=float(cut.detach()), reward=float(reward), queue=queue.cpu().numpy(), counts=counts.cpu().numpy(), grad=grad)


HYBRID = ["hybrid_short", "hybrid_p2", "hybrid_p3", "hybrid_l10", "hybrid_n2", "hybrid_4x4", "hybrid_half", "hybrid_l30", "hybrid_5x5",
          "hybrid_n2l30", "sweep_g"]


@pytest.mark.parametrize("name", HYBRID)
def test_stepwise_hybrid_matches_reference(cuda, golden_dir, name):
    """Reference runs of hybrid networks -- the fused kernels' goldens (3 x 3 grids of 5 / 10 m lanes, two lanes per approach, 4 x 4,
    BASELINE config 4's 600-step episode) and three the fused kernels cannot or can only just hold (30 m lanes: 720 cells + lanes;
    5 x 5 intersections: 144 IDM lanes; two lanes per approach with 30 m lanes: 252 lanes + 1 296 cells): vehicle counts, queues
    <= 1e-5, reward <= 1e-5, the whole d reward / d action <= 1e-4 and the gradient of the reward restricted to its first t0 steps."""
    path = os.path.join(golden_dir, "itscp_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path)
    net, t, m = _net(cuda, g)
    o = _run(cuda, net, m, g["action"])
    assert o["counts"][0] == m["n_vehicle_spawned"]
    assert state_report("queues vs reference", o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    scale = np.abs(g["g_action"]).max()
    err = np.abs(o["grad"] - g["g_action"]).max() / scale
    print("%s: %d lanes, %d cells, %d IDM lanes, %d vehicles, %d deposits, %d events; gradient error / max|g| %.2e" % (
        name, t.n_lanes, t.n_cells, net.n_micro, o["counts"][0], o["counts"][1], o["counts"][2], err))
    assert err <= TOL_GRAD
    for t0, ref in zip(g["g_action_cut_steps"], g["g_action_cut"]) if "g_action_cut_steps" in g.files else []:
        if name == "hybrid" and t0 > 510:
            continue
        oc = _run(cuda, net, m, g["action"], loss_steps=int(t0))
        assert np.abs(oc["grad"] - ref).max() <= TOL_GRAD * scale, int(t0)
    o2 = _run(cuda, net, m, g["action"])
    assert np.array_equal(o2["grad"], o["grad"]) and np.array_equal(o2["queue"], o["queue"]) and o2["reward"] == o["reward"]     # repeatable


@pytest.mark.parametrize("name", ["macro_small", "macro", "macro_2x2", "macro_half", "macro_long", "macro_3x3x3", "sweep_b"])
def test_stepwise_macro_matches_reference(cuda, golden_dir, name):
    """Macro-only networks (no IDM lane, no hand-off): the reference's macro runs, incl. 106 200 loss samples (the running mean's window
    slides) and the 360-lane network beyond one workgroup."""
    path = os.path.join(golden_dir, "itscp_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path)
    net, t, m = _net(cuda, g)
    o = _run(cuda, net, m, g["action"])
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert rel_max(o["grad"], g["g_action"]) <= TOL_GRAD
    assert o["counts"][0] == 0 and o["counts"][2] == 0


@pytest.mark.parametrize("name", ["micro_small", "micro", "micro_2x2", "micro_p2", "micro_l10", "micro_jam_a", "micro_jam_b", "micro_jam_c", "sweep_c", "sweep_e"])
def test_stepwise_micro_mode_matches_reference(cuda, golden_dir, name):
    """itscp `micro` mode: every lane an IDM lane, source lanes admit waiting vehicles against the host's recorded draws; 16, 40 and
    112 lanes (the last beyond the fused kernels' 64).  The reference steps these lanes with the autodiff MicroLane in float32 tensor
    arithmetic; the kernels follow that ladder in this mode (csrc/idm_device.hpp idm_step_f32): queues <= 1e-5 like every other mode
    (measured 1.7e-7 ... 2.2e-6)."""
    from dhts.stepwise import StepwiseNetwork
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m, rows = itscp_micro_tables(g)
    net = StepwiseNetwork(t, rows, cuda, lane_capacity=32)
    o = _run(cuda, net, m, g["action"])
    assert o["counts"][0] == m["n_vehicle_spawned"] and o["counts"][3] == len(g["rand_draws"])
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert rel_max(o["grad"], g["g_action"]) <= TOL_GRAD


@pytest.mark.parametrize("name", ["eval_macro", "eval_macro_2x2", "eval_hybrid_short", "eval_hybrid_p2", "eval_hybrid", "eval_hybrid_4x4",
                                  "eval_macro_3x3x3", "eval_hybrid_n2l30", "eval_hybrid_5x5", "sweep_a", "sweep_f"])
def test_stepwise_evaluation_episode_matches_reference(cuda, golden_dir, name):
    """ItscpEnv.step(action, False) (Trainer.evaluate): hard thresholds."""
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    net, t, m = _net(cuda, g)
    o = _run(cuda, net, m, g["action"], differentiable=False)
    assert o["counts"][0] == m["n_vehicle_spawned"]
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))


@pytest.mark.parametrize("name, seed", [("hybrid_p2", 3), ("hybrid_l10", 4), ("hybrid_5x5", 5)])
def test_stepwise_random_actions_vs_oracle_and_fused(cuda, golden_dir, oracle, name, seed):
    """Other signal schedules (other spawn times and lane orders) against the CPU oracle; where the fused kernels hold the network,
    against them as well (same events, queues and gradient to rounding)."""
    import torch
    from dhts import ops
    from dhts.network import group_routes
    from dhts.stepwise import StepwiseNetwork
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m = itscp_hybrid_tables(g)
    routes = np.concatenate([g["spawn_routes"]] * 4)
    gr, ptr = group_routes(routes, t.n_lanes)
    net = StepwiseNetwork(t, routes, cuda)
    rng = np.random.default_rng(seed)
    try:
        t.check_kernel_limits()
        fused = ops.DeviceHybridTables(t, routes, cuda)
    except ValueError:
        fused = None
    worst = 0.0
    for k in range(4):
        act = rng.uniform(0.1, 0.9, len(g["action"])).astype(np.float32)
        o = _run(cuda, net, m, act)
        ref = oracle.net_hybrid(t, gr, ptr, act, *_args(m))
        assert ref["rc"] == 0 and (o["counts"][0], o["counts"][1]) == (ref["n_spawned"], ref["n_deposits"]), k
        assert state_report("queues vs oracle (%d)" % k, o["queue"], ref["queue"]) <= TOL_STATE
        assert abs(o["reward"] - ref["reward"]) <= 1e-5 * abs(ref["reward"])
        # an entry whose value the ORACLE itself does not reproduce after a one-ulp nudge of the action is float32 noise (a solver
        # branch decided by the last bit, amplified over a standing queue: DESIGN section 2): such entries are not compared
        nudged = oracle.net_hybrid(t, gr, ptr, np.nextafter(act, np.float32(1.0)), *_args(m))["g_action"]
        scale = np.abs(ref["g_action"]).max()
        stable = np.abs(nudged - ref["g_action"]) <= 2e-5 * scale
        assert stable.sum() >= len(act) - 2, (k, int((~stable).sum()))
        worst = max(worst, (np.abs(o["grad"] - ref["g_action"])[stable]).max() / scale)
        if fused is not None:
            a = torch.tensor(act[None, :], device=cuda, requires_grad=True)
            cut, reward, queue, counts = ops.net_hybrid_rollout(a, fused, *_args(m))
            cut.sum().backward()
            assert (int(counts[0, 0]), int(counts[0, 1])) == (o["counts"][0], o["counts"][1])
            assert rel_max(o["queue"], queue[0].cpu().numpy()) <= 1e-6
            assert rel_max(o["grad"], a.grad[0].cpu().numpy()) <= 1e-5
    assert worst <= 0.2 * TOL_GRAD, worst


def test_stepwise_lane_capacity_fault_is_loud(cuda, golden_dir):
    """A micro lane that would hold more vehicles than the launch was sized for: DHTS_FAULT_CAPACITY -> dhts.ops.CapacityError."""
    from dhts import ops
    g = np.load(os.path.join(golden_dir, "itscp_micro.npz"))
    from dhts.stepwise import StepwiseNetwork
    t, m, rows = itscp_micro_tables(g)
    net = StepwiseNetwork(t, rows, cuda, lane_capacity=1)
    with pytest.raises(ops.CapacityError):
        _run(cuda, net, m, g["action"])


def test_stepwise_episode_time(cuda, golden_dir):
    """The review's bar: a differentiable episode of a network beyond the fused limits in <= 0.3 s (it ran for minutes lane by lane)."""
    import torch
    path = os.path.join(golden_dir, "itscp_hybrid_n2l30.npz")
    if not os.path.exists(path):
        path = os.path.join(golden_dir, "itscp_hybrid_5x5.npz")
    g = np.load(path)
    net, t, m = _net(cuda, g)
    _run(cuda, net, m, g["action"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        _run(cuda, net, m, g["action"])
    torch.cuda.synchronize()
    dt_ = (time.perf_counter() - t0) / 3
    print("stepwise differentiable episode (%s: %d lanes, %d cells, %d steps): %.1f ms" % (os.path.basename(path), t.n_lanes, t.n_cells, t.T, 1e3 * dt_))
    assert dt_ <= 0.3


@pytest.mark.parametrize("name", ["hybrid_n2l30", "hybrid_5x5", "micro_2x2", "macro_3x3x3"])
def test_env_step_takes_the_stepwise_path_beyond_the_fused_limits(cuda, golden_dir, name):
    """ItscpEnv.step(action, True) -- the reference's entry point (trainer.py:172-190) -- on networks the fused kernels cannot hold
    (252 lanes + 1 152 cells; 144 IDM lanes; 112 IDM lanes in `micro` mode; 360 lanes + 2 124 cells in `macro` mode): the episode
    runs on the stepwise device path (persistent form) and reproduces the reference's run; an evaluation episode runs there too."""
    import torch
    from test_itscp_gpu import build_env
    from test_oracle_golden import meta_of
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    m = meta_of(g)
    micro = name.startswith("micro")
    env = build_env(g, m, replay_routes=not micro)
    if name.startswith("hybrid"):
        env.fused_routes = g["spawn_routes"]
    if micro:
        env.fused_draws = g["rand_draws"]
    keys = list(env.lane.keys())
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    t0 = time.perf_counter()
    obs, reward, done, info = env.step(action, True)
    reward.backward()
    grad = action.grad.cpu().numpy()
    t1 = time.perf_counter()
    assert env._fused_cache[0] == "stepwise" and env._fused_done and env.last_path == "stepwise" and env._fused_cache[1].persistent
    queue = np.array([env.queue_length[k] for k in keys])
    tol_q = TOL_STATE
    assert state_report("env.step %s: queues vs reference" % name, queue, g["queue"]) <= tol_q
    assert abs(float(reward.detach()) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert np.abs(grad - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()
    if not name.startswith("macro"):
        assert env.fused_counts[0] == m["n_vehicle_spawned"]
    print("env.step %s on the stepwise path: %.1f ms (first call: tables built and uploaded)" % (name, 1e3 * (t1 - t0)))
    # an evaluation episode on a twin (Trainer.evaluate)
    env.rewind()
    twin = env.episode_copy()
    if micro:
        twin.fused_draws = g["rand_draws"]
    with torch.no_grad():
        _, r_eval, _, _ = twin.step(action.detach(), False)
    assert twin.last_path == "stepwise" and np.isfinite(float(r_eval))


# ---- the PERSISTENT form: one kernel per direction, one workgroup per replica (same device functions) -------------------------------
@pytest.mark.parametrize("name", HYBRID + ["macro_small", "macro", "macro_2x2", "macro_long", "macro_3x3x3"])
def test_persistent_form_equals_stepwise_form_and_reference(cuda, golden_dir, name):
    """Every hybrid / macro golden through the persistent kernels: the reference's queues, reward, vehicle count and gradient, and the
    SAME numbers as the stepwise form (same device functions, the lanes' steps item by item with the operator's operations: queues
    bit-identical, gradient to rounding of its accumulation order)."""
    path = os.path.join(golden_dir, "itscp_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path)
    net_s, t, m = _net(cuda, g)
    net_p, _, _ = _net(cuda, g, persistent=True)
    o_s = _run(cuda, net_s, m, g["action"])
    o_p = _run(cuda, net_p, m, g["action"])
    assert np.array_equal(o_p["counts"], o_s["counts"]) and o_p["counts"][0] == m["n_vehicle_spawned"]
    assert state_report("queues vs reference", o_p["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o_p["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    scale = np.abs(g["g_action"]).max()
    assert np.abs(o_p["grad"] - g["g_action"]).max() <= TOL_GRAD * scale
    assert np.array_equal(o_p["queue"], o_s["queue"]) and o_p["reward"] == o_s["reward"]
    print("%s: persistent vs stepwise gradient %.1e of max|g|" % (name, np.abs(o_p["grad"] - o_s["grad"]).max() / scale))
    assert np.abs(o_p["grad"] - o_s["grad"]).max() <= 1e-6 * scale
    for t0, ref in zip(g["g_action_cut_steps"][:1], g["g_action_cut"][:1]) if "g_action_cut_steps" in g.files else []:
        oc = _run(cuda, net_p, m, g["action"], loss_steps=int(t0))
        assert np.abs(oc["grad"] - ref).max() <= TOL_GRAD * scale, int(t0)
    o2 = _run(cuda, net_p, m, g["action"])
    assert np.array_equal(o2["grad"], o_p["grad"]) and np.array_equal(o2["queue"], o_p["queue"])            # repeatable
    oe_s = _run(cuda, net_s, m, g["action"], differentiable=False)
    oe_p = _run(cuda, net_p, m, g["action"], differentiable=False)
    assert np.array_equal(oe_p["queue"], oe_s["queue"]) and np.array_equal(oe_p["counts"], oe_s["counts"])     # evaluation episodes too


@pytest.mark.parametrize("name", ["micro_small", "micro_2x2", "micro_p2", "micro_l10", "micro_jam_a", "micro_jam_b", "micro_jam_c", "sweep_c", "sweep_e"])
def test_persistent_form_micro_mode(cuda, golden_dir, name):
    from dhts.stepwise import StepwiseNetwork
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m, rows = itscp_micro_tables(g)
    o = _run(cuda, StepwiseNetwork(t, rows, cuda, persistent=True), m, g["action"])
    o_s = _run(cuda, StepwiseNetwork(t, rows, cuda), m, g["action"])
    assert o["counts"][0] == m["n_vehicle_spawned"] and o["counts"][3] == len(g["rand_draws"])
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE and rel_max(o["grad"], g["g_action"]) <= TOL_GRAD
    assert np.array_equal(o["queue"], o_s["queue"]) and np.abs(o["grad"] - o_s["grad"]).max() <= 1e-6 * np.abs(o_s["grad"]).max()


def test_persistent_replicas_equal_their_single_runs(cuda, golden_dir):
    """R replicas of a network beyond the fused limits (own inflow schedules each) as R workgroups of one launch pair: every replica
    equals its own single-replica run bit for bit; time per batch printed."""
    import copy
    import torch
    from dhts.stepwise import StepwiseNetwork
    name = "hybrid_n2l30" if os.path.exists(os.path.join(golden_dir, "itscp_hybrid_n2l30.npz")) else "hybrid_5x5"
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m = itscp_hybrid_tables(g)
    rng = np.random.default_rng(3)
    R = 4
    tabs, acts = [], []
    for r in range(R):
        x = copy.copy(t)
        x.schedule = np.ascontiguousarray(t.schedule * (1.0 if r == 0 else rng.uniform(0.5, 1.0)))
        tabs.append(x)
        acts.append(g["action"] if r == 0 else rng.uniform(0.1, 0.9, len(g["action"])).astype(np.float32))
    net = StepwiseNetwork(tabs, g["spawn_routes"], cuda, persistent=True)
    a = torch.tensor(np.stack(acts), device=cuda, requires_grad=True)
    cut, reward, queue, counts = net.rollout(a, *_args(m))
    cut.sum().backward()
    G, Q = a.grad.cpu().numpy(), queue.cpu().numpy()
    assert abs(float(reward[0]) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"])) and int(counts[0, 0]) == m["n_vehicle_spawned"]
    assert not np.array_equal(Q[0], Q[1])
    for r in range(R):
        o = _run(cuda, StepwiseNetwork(tabs[r], g["spawn_routes"], cuda, persistent=True), m, acts[r])
        assert np.array_equal(o["queue"], Q[r]) and np.array_equal(o["grad"], G[r]) and o["reward"] == float(reward[r])
    for R2 in (1, 64):
        tb = [tabs[r % R] for r in range(R2)]
        net2 = StepwiseNetwork(tb, g["spawn_routes"], cuda, persistent=True)
        a2 = torch.tensor(np.stack([acts[r % R] for r in range(R2)]), device=cuda, requires_grad=True)

        def ep():
            a2.grad = None
            c2, _, _, _ = net2.rollout(a2, *_args(m), check_faults=False)
            c2.sum().backward()
        ep(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ep()
        torch.cuda.synchronize()
        print("persistent form, %s (%d lanes, %d cells, %d steps), %2d replica(s): %.2f ms per differentiable batch episode" % (
            name, t.n_lanes, t.n_cells, t.T, R2, 1e3 * (time.perf_counter() - t0) / 3))


@pytest.mark.parametrize("name", ["hybrid_p2", "hybrid_l30", "hybrid_n2l30", "hybrid_5x5"])
def test_persistent_form_at_the_geometric_capacity(cuda, golden_dir, name):
    """The capacity sized from the geometry (dhts.stepwise.default_lane_capacity: what ItscpEnv and the trainer's replica batches start
    from) keeps the micro side's running state in LDS -- other instantiations of the persistent kernels than at 32 slots per lane.
    Capacity is storage only: same queues bit for bit, same gradient, and the reference's."""
    from dhts.stepwise import default_lane_capacity
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m = itscp_hybrid_tables(g)
    cap = default_lane_capacity(t, m["vehicle_length"])
    assert cap in (4, 8, 16, 32)
    net_c, _, _ = _net(cuda, g, lane_capacity=cap, persistent=True)
    net_32, _, _ = _net(cuda, g, lane_capacity=32, persistent=True)
    net_s, _, _ = _net(cuda, g, lane_capacity=cap, persistent=False)
    o_c, o_32, o_s = (_run(cuda, n, m, g["action"]) for n in (net_c, net_32, net_s))
    print("%s: capacity %d" % (name, cap))
    assert np.array_equal(o_c["queue"], o_32["queue"]) and np.array_equal(o_c["counts"], o_32["counts"]) and o_c["reward"] == o_32["reward"]
    assert np.array_equal(o_c["queue"], o_s["queue"])
    scale = np.abs(g["g_action"]).max()
    assert np.abs(o_c["grad"] - o_32["grad"]).max() <= 1e-6 * scale and np.abs(o_c["grad"] - o_s["grad"]).max() <= 1e-6 * scale
    assert np.abs(o_c["grad"] - g["g_action"]).max() <= TOL_GRAD * scale
    assert state_report("queues vs reference", o_c["queue"].T, g["queue"]) <= TOL_STATE
    oe_c = _run(cuda, net_c, m, g["action"], differentiable=False)
    oe_32 = _run(cuda, net_32, m, g["action"], differentiable=False)
    assert np.array_equal(oe_c["queue"], oe_32["queue"])


def test_env_capacity_ladder_starts_at_the_geometric_capacity(cuda, golden_dir):
    """ItscpEnv on a network beyond the fused limits starts the stepwise path at the geometry's capacity; an episode that outgrows a
    capacity (forced here: one slot per lane) is retried at the next rung with the same routes -- and gives the reference's numbers."""
    import torch
    from test_itscp_gpu import build_env
    from test_oracle_golden import meta_of
    from dhts.stepwise import default_lane_capacity
    g = np.load(os.path.join(golden_dir, "itscp_hybrid_n2l30.npz"))
    m = meta_of(g)
    env = build_env(g, m, replay_routes=True)
    env.fused_routes = g["spawn_routes"]
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    obs, reward, done, info = env.step(action, True)
    t, _ = itscp_hybrid_tables(g)
    assert env.last_path == "stepwise" and env._fused_cache[1].lane_capacity == default_lane_capacity(t, m["vehicle_length"]) == 8
    r0 = float(reward.detach())
    assert abs(r0 - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    env2 = build_env(g, m, replay_routes=True)
    env2.fused_routes = g["spawn_routes"]
    env2._fused_lane_capacity = 1
    env2._fused_prefer_stepwise = True
    a2 = torch.tensor(g["action"], device=cuda, requires_grad=True)
    _, reward2, _, _ = env2.step(a2, True)
    reward2.backward()
    assert env2.last_path == "stepwise" and env2._fused_cache[1].lane_capacity == 32
    assert float(reward2.detach()) == r0
    assert np.abs(a2.grad.cpu().numpy() - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()


@pytest.mark.parametrize("name", ["hybrid_n2l30", "hybrid_5x5", "macro_3x3x3", "micro_2x2"])
def test_persistent_instantiations_agree(cuda, golden_dir, name):
    """The persistent kernels come in instantiations by what a workgroup's LDS holds (static tables + rows + ghosts, the micro side's
    running state, state rows / cotangent planes; csrc/netstep_hybrid.hip ns_plan).  DHTS_OPT_NETSTEP_LDS_KB shrinks the budget, so that
    the same episode runs through the others -- down to nothing staged: same queues and counts bit for bit, same gradient."""
    from dhts import _lib
    from dhts.stepwise import StepwiseNetwork, default_lane_capacity
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    if name.startswith("micro"):
        t, m, routes = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        routes = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
    net = StepwiseNetwork(t, routes, cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=True)
    lib = _lib.lib()
    try:
        ref = _run(cuda, net, m, g["action"])
        ref_e = _run(cuda, net, m, g["action"], differentiable=False)
        scale = np.abs(ref["grad"]).max()
        assert np.abs(ref["grad"] - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()
        for kb in (120, 96, 72, 56, 40, 24, 8):
            assert lib.dhts_set_option(_lib.OPT_NETSTEP_LDS_KB, kb) == 0
            o = _run(cuda, net, m, g["action"])
            assert np.array_equal(o["queue"], ref["queue"]) and np.array_equal(o["counts"], ref["counts"]) and o["reward"] == ref["reward"], kb
            assert np.abs(o["grad"] - ref["grad"]).max() <= 1e-6 * scale, kb
            oe = _run(cuda, net, m, g["action"], differentiable=False)
            assert np.array_equal(oe["queue"], ref_e["queue"]), kb
        lib.dhts_set_option(_lib.OPT_NETSTEP_LDS_KB, 0)
        # fewer threads per workgroup (DHTS_OPT_NETSTEP_BLOCK): the ordered prefix sums split differently over the wavefronts, so the
        # running means may differ in their last float64 bits -- queues to 1e-6 instead of bit for bit
        for block in (512, 256):
            assert lib.dhts_set_option(_lib.OPT_NETSTEP_BLOCK, block) == 0
            o = _run(cuda, net, m, g["action"])
            assert np.array_equal(o["counts"], ref["counts"]) and rel_max(o["queue"], ref["queue"]) <= 1e-6, block
            assert np.abs(o["grad"] - ref["grad"]).max() <= 1e-5 * scale, block
    finally:
        lib.dhts_set_option(_lib.OPT_NETSTEP_LDS_KB, 0)
        lib.dhts_set_option(_lib.OPT_NETSTEP_BLOCK, 0)


@pytest.mark.parametrize("name", ["hybrid_n2l30", "hybrid_p2"])
def test_lane_capacity_is_storage_only(cuda, golden_dir, name):
    """Any capacity the episode fits in gives the same numbers (odd ones too: the loops over vehicle slots address `capacity` slots
    per lane but visit a power of two per lane, or all of them); one it does not fit in is a loud CapacityError, never a wrong number."""
    from dhts import ops
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    ref = None
    fitted = []
    for cap in (32, 7, 6, 5, 4, 3, 2):
        for persistent in (True, False):
            net, _, m = _net(cuda, g, lane_capacity=cap, persistent=persistent)
            try:
                o = _run(cuda, net, m, g["action"])
            except ops.CapacityError:
                continue
            fitted.append((cap, persistent))
            if ref is None:
                ref = o
                assert np.abs(o["grad"] - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()
            assert np.array_equal(o["queue"], ref["queue"]) and np.array_equal(o["counts"], ref["counts"]), (cap, persistent)
            assert np.abs(o["grad"] - ref["grad"]).max() <= 1e-6 * np.abs(ref["grad"]).max(), (cap, persistent)
    print("%s: capacities that fit:" % name, fitted)
    assert (32, True) in fitted and (32, False) in fitted and len(fitted) >= 6


def test_network_wider_than_the_workgroup(cuda, oracle):
    """A 9 x 9 hybrid grid built by the environment itself (1 296 lanes -- more than the persistent kernels' 1 024 threads, so every
    item loop takes several rounds and the static tables do not fit LDS: the unstaged instantiation; ~470 IDM lanes): persistent
    form = stepwise form bit for bit, and both against the CPU oracle."""
    import torch
    from dhts.network import HybridNetworkTables, group_routes
    from dhts.stepwise import StepwiseNetwork, default_lane_capacity
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp import problem as problems
    np.random.seed(5)
    env = ItscpEnv()
    env.schedule_callback = problems.problem_1
    for k, v in dict(num_intersection=9, lane_length=5.0, num_lane=1, policy_length=2, signal_length=1, mode="hybrid", speed_limit=60.0,
                     random_seed=5).items():
        env.config[k] = v
    env.reset()
    t = HybridNetworkTables.from_env(env)
    assert t.n_lanes > 1024
    spawn = [l for l in range(t.n_lanes) if t.lane_macro[l] == 0 and any(t.lane_macro[a] for a in t.prev_lanes[l])]
    routes = [list(env.simulator.create_random_route(l).route)[:32] for l in spawn for _ in range(4)]
    routes = np.asarray([r + [-1] * (32 - len(r)) for r in routes], dtype=np.int32)
    gr, ptr = group_routes(routes, t.n_lanes)
    args = (env.config["num_intersection"] ** 2, env.config["simulation_frequency"] * env.config["signal_length"],
            1.0 / env.config["simulation_frequency"], env.simulator.speed_limit, env.config["static_speed"], env.simulator.vehicle_length)
    cap = default_lane_capacity(t, env.simulator.vehicle_length)
    n_action = env.config["policy_length"] * env.config["num_intersection"] ** 2
    act = np.random.default_rng(1).uniform(0.1, 0.9, n_action).astype(np.float32)
    outs = []
    for persistent in (True, False):
        net = StepwiseNetwork(t, routes, cuda, lane_capacity=cap, persistent=persistent)
        a = torch.tensor(act, device=cuda, requires_grad=True)
        t0 = time.perf_counter()
        cut, reward, queue, counts = net.rollout(a, *args)
        cut.backward()
        torch.cuda.synchronize()
        outs.append(dict(queue=queue.cpu().numpy(), grad=a.grad.cpu().numpy(), counts=counts.cpu().numpy(), reward=float(reward)))
        print("%d lanes, %d cells, %d IDM lanes, %d steps, %s form: %.1f ms (first call)" %
              (t.n_lanes, t.n_cells, net.n_micro, t.T, "persistent" if persistent else "stepwise", 1e3 * (time.perf_counter() - t0)))
    p, s = outs
    assert np.array_equal(p["queue"], s["queue"]) and np.array_equal(p["counts"], s["counts"]) and p["reward"] == s["reward"]
    scale = np.abs(s["grad"]).max()
    assert np.abs(p["grad"] - s["grad"]).max() <= 1e-6 * scale
    ref = oracle.net_hybrid(t, gr, ptr, act, *args)
    assert ref["rc"] == 0 and (p["counts"][0], p["counts"][1]) == (ref["n_spawned"], ref["n_deposits"])
    assert state_report("queues vs oracle", p["queue"], ref["queue"]) <= TOL_STATE
    assert abs(p["reward"] - ref["reward"]) <= 1e-5 * abs(ref["reward"])
    assert np.abs(p["grad"] - ref["g_action"]).max() <= TOL_GRAD * np.abs(ref["g_action"]).max()
    # the environment itself routes a single episode of this size to the stepwise FORM (dhts.stepwise.persistent_form_pays)
    a_env = torch.tensor(act, device=cuda, requires_grad=True)
    _, r_env, _, _ = env.step(a_env, True)
    r_env.backward()
    assert env.last_path == "stepwise" and not env._fused_cache[1].persistent
    assert np.isfinite(float(r_env.detach())) and np.isfinite(a_env.grad.cpu().numpy()).all() and float(a_env.grad.abs().max()) > 0


def test_env_ladder_survives_a_fused_launch_that_does_not_fit(cuda, golden_dir, monkeypatch):
    """a fused launch whose LDS staging does not fit comes back as DhtsError (not CapacityError).  The ladder treats it
    like an exceeded capacity: next rung (the stepwise path), same routes, the reference's numbers.  (64 IDM lanes at 128 vehicle slots
    DO fit on gfx950 -- first half of the test --, so the refusal is injected for the second.)"""
    import torch
    from dhts import _lib, ops
    from test_itscp_gpu import build_env
    from test_oracle_golden import meta_of
    g = np.load(os.path.join(golden_dir, "itscp_hybrid_4x4.npz"))
    m = meta_of(g)

    def episode():
        env = build_env(g, m, replay_routes=True)
        env.fused_routes = g["spawn_routes"]
        env._fused_lane_capacity = 128
        action = torch.tensor(g["action"], device=cuda, requires_grad=True)
        _, reward, _, _ = env.step(action, True)
        reward.backward()
        assert env._fused_done
        assert abs(float(reward.detach()) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
        assert np.abs(action.grad.cpu().numpy() - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()
        assert env.fused_counts[0] == m["n_vehicle_spawned"]
        return env
    env = episode()
    assert env.last_path == "fused" and env._fused_cache[1].lane_capacity == 128
    real = ops.net_hybrid_rollout

    def refuse(a, tab, *args, **kw):
        if tab.lane_capacity == 128:
            e = _lib.DhtsError("dhts_net_hybrid_rollout_fwd: DHTS_E_INVALID (injected: LDS staging does not fit)")
            e.status = _lib.E_INVALID
            raise e
        return real(a, tab, *args, **kw)
    monkeypatch.setattr(ops, "net_hybrid_rollout", refuse)
    env = episode()
    assert env.last_path == "stepwise" and env._fused_cache[1].lane_capacity == 32


def test_env_does_not_swallow_library_errors(cuda, golden_dir, monkeypatch):
    """only a DHTS_E_INVALID sizing refusal has another way to run.  A failed launch (or any error of the macro paths)
    reaches the caller instead of turning into four silent retries and a lane-by-lane episode."""
    import torch
    from dhts import _lib, ops
    from test_itscp_gpu import build_env
    from test_oracle_golden import meta_of
    g = np.load(os.path.join(golden_dir, "itscp_hybrid_short.npz"))
    env = build_env(g, meta_of(g), replay_routes=True)
    env.fused_routes = g["spawn_routes"]

    def broken(*a, **kw):
        e = _lib.DhtsError("dhts_net_hybrid_rollout_fwd failed: DHTS_E_LAUNCH (HIP launch failed)")
        e.status = _lib.E_LAUNCH
        raise e
    monkeypatch.setattr(ops, "net_hybrid_rollout", broken)
    with pytest.raises(_lib.DhtsError, match="E_LAUNCH"):
        env.step(torch.tensor(g["action"], device=cuda, requires_grad=True), True)
    assert not getattr(env, "fused_overflowed", False)


def test_event_list_overflow_is_told_apart_from_a_full_lane(cuda, golden_dir):
    """the hand-off event list (dhts_netstep_tables::max_events) is sized from the network, an episode that needs
    more comes back as DHTS_FAULT_CAPACITY with index -2, and ItscpEnv retries with a larger LIST (not with more vehicle slots per
    lane, which cannot help) -- same episode, the reference's numbers."""
    import torch
    from dhts import ops
    from dhts.stepwise import StepwiseNetwork
    from test_itscp_gpu import build_env
    from test_oracle_golden import meta_of
    g = np.load(os.path.join(golden_dir, "itscp_hybrid_p2.npz"))
    t, m = itscp_hybrid_tables(g)
    a = torch.tensor(g["action"], device=cuda, requires_grad=True)
    for persistent in (False, True):
        small = StepwiseNetwork(t, g["spawn_routes"], cuda, lane_capacity=32, max_events=4, persistent=persistent)
        with pytest.raises(ops.CapacityError) as ei:
            small.rollout(a, *_args(m))
        assert ei.value.index == -2
    # the environment: forced onto the stepwise path with a list of four events -> one retry with the hard bound, same numbers
    env = build_env(g, m, replay_routes=True)
    env.fused_routes = g["spawn_routes"]
    env._fused_prefer_stepwise, env._fused_lane_capacity, env._stepwise_max_events = True, 32, 4
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    _, reward, _, _ = env.step(action, True)
    reward.backward()
    assert env.last_path == "stepwise" and env._fused_cache[1].lane_capacity == 32          # the lane capacity did not climb
    assert env._fused_cache[1].max_events == m["T"] * (4 * env._fused_cache[1].n_micro + 2 * env._fused_cache[1].n_caps) + 64
    assert abs(float(reward.detach()) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert np.abs(action.grad.cpu().numpy() - g["g_action"]).max() <= TOL_GRAD * np.abs(g["g_action"]).max()


@pytest.mark.parametrize("form", ["fused", "stepwise", "persistent"])
def test_tensor_ladder_is_said_by_the_host_not_inferred_from_source_lanes(cuda, golden_dir, oracle, form):
    """dhts_hybrid_tables::micro_tensor_ladder.  A network with an IDM source lane whose lanes are dMicroLane objects
    (flag 0) steps in the analytic operator's float64 ladder -- oracle and every device form agree on it, and it is NOT the float32
    tensor ladder of itscp `micro` mode (flag 1), which the same forms reproduce from the reference's fixture."""
    import torch
    from dhts import ops
    from dhts.network import group_routes
    from dhts.stepwise import StepwiseNetwork
    from test_oracle_golden import itscp_micro_tables
    g = np.load(os.path.join(golden_dir, "itscp_micro_small.npz"))
    res = {}
    for flag in (True, False):
        t, m, routes = itscp_micro_tables(g)
        t.set_micro_sources(g["rand_draws"], tensor_ladder=flag)
        gr, ptr = group_routes(routes, t.n_lanes)
        o = oracle.net_hybrid(t, gr, ptr, g["action"], *_args(m))
        a = torch.tensor(g["action"], device=cuda, requires_grad=True)
        if form == "fused":
            cut, reward, queue, counts = ops.net_hybrid_rollout(a[None], ops.DeviceHybridTables(t, routes, cuda), *_args(m))
            cut.sum().backward()
            q, grad = queue[0].cpu().numpy(), a.grad.cpu().numpy()
        else:
            net = StepwiseNetwork(t, routes, cuda, lane_capacity=32, persistent=form == "persistent")
            cut, reward, queue, counts = net.rollout(a, *_args(m))
            cut.backward()
            q, grad = queue.cpu().numpy(), a.grad.cpu().numpy()
        assert o["rc"] == 0 and rel_max(q, o["queue"]) <= TOL_STATE
        assert np.abs(grad - o["g_action"]).max() <= TOL_GRAD * np.abs(o["g_action"]).max()
        res[flag] = q
    assert rel_max(res[True].T, g["queue"]) <= TOL_STATE                      # flag 1 = the reference's `micro` mode
    assert not np.array_equal(res[True], res[False])                          # ... and the two ladders are different arithmetic


@pytest.mark.parametrize("persistent", [False, True])
def test_several_rollouts_before_one_backward(cuda, golden_dir, persistent):
    """every differentiable rollout owns its workspace and keeps the tables it was stepped with, so episodes can be
    summed before ONE backward() (Trainer.train_epoch with num_episode_per_epoch > 1) -- here two episodes of a `micro` mode
    network with different admission draws and actions: the summed gradient equals the two separate ones."""
    import torch
    from dhts.stepwise import StepwiseNetwork
    from test_oracle_golden import itscp_micro_tables
    g = np.load(os.path.join(golden_dir, "itscp_micro_small.npz"))
    t, m, routes = itscp_micro_tables(g)
    rng = np.random.default_rng(2)
    n = m["T"] * int(t.lane_source.sum())                                # one draw per source lane and step at most
    d1 = np.concatenate([np.asarray(g["rand_draws"], dtype=np.float64), np.full(n, 2.0)])[:n]      # the recorded stream (the rest is never drawn)
    d2 = rng.random(n)
    t.set_micro_sources(d1)
    net = StepwiseNetwork(t, routes, cuda, lane_capacity=32, persistent=persistent)
    a1 = torch.tensor(g["action"], device=cuda, requires_grad=True)
    a2 = torch.tensor(rng.uniform(0.2, 0.8, len(g["action"])).astype(np.float32), device=cuda, requires_grad=True)
    sep = []
    for d, a in ((d1, a1), (d2, a2)):
        net.set_draws(d)
        cut, *_ = net.rollout(a, *_args(m))
        cut.backward()
        sep.append((float(cut.detach()), a.grad.clone()))
        a.grad = None
    net.set_draws(d1)
    c1, *_ = net.rollout(a1, *_args(m))
    net.set_draws(d2)
    c2, *_ = net.rollout(a2, *_args(m))
    net.rollout(a2, *_args(m), differentiable=False)                    # (an evaluation episode in between takes the shared workspace)
    (c1 + c2).backward()
    assert float(c1.detach()) == sep[0][0] and float(c2.detach()) == sep[1][0]
    assert torch.equal(a1.grad, sep[0][1]) and torch.equal(a2.grad, sep[1][1])
    assert abs(sep[0][0] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))


@pytest.mark.parametrize("name", ["eval_micro_small", "eval_micro", "eval_micro_2x2", "sweep_d"])
def test_micro_mode_evaluation_episode_matches_reference(cuda, golden_dir, name):
    """Evaluation episodes in `micro` mode (Trainer.evaluate on run_itscp_micro.sh's environment): the reference holds Python floats all
    the way there, the kernels step the lanes in the analytic operator's float64 ladder (the float32 tensor ladder is for differentiable
    episodes only).  Stepwise form, persistent form and the fused evaluation kernel against the reference's run."""
    import torch
    from dhts import ops
    from dhts.stepwise import StepwiseNetwork, default_lane_capacity
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m, rows = itscp_micro_tables(g)
    for persistent in (False, True):
        net = StepwiseNetwork(t, rows, cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=persistent)
        o = _run(cuda, net, m, g["action"], differentiable=False)
        assert o["counts"][0] == m["n_vehicle_spawned"] and o["counts"][3] == len(g["rand_draws"])
        assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE, persistent
        assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    try:
        dtab = ops.DeviceHybridTables(t, rows, cuda)
    except ValueError:                  # (112 / 160 IDM lanes: beyond the fused kernels)
        assert name in ("eval_micro_2x2", "sweep_d")
        return
    reward, queue, counts = ops.net_hybrid_eval(torch.tensor(g["action"][None], device=cuda), dtab, *_args(m))
    assert int(counts[0, 0]) == m["n_vehicle_spawned"]
    assert rel_max(queue[0].cpu().numpy().T, g["queue"]) <= TOL_STATE
    assert abs(float(reward[0]) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))


@pytest.mark.parametrize("name, seed", [("micro_small", 11), ("micro_p2", 12), ("micro_l10", 13)])
def test_micro_mode_random_actions_vs_oracle(cuda, golden_dir, oracle, name, seed):
    """Other signal schedules in `micro` mode (other admissions, other lane changes) on the float32 tensor ladder: the persistent kernels
    against the CPU oracle -- same draws consumed, same vehicles, queues, reward, gradient."""
    from dhts.network import group_routes
    from dhts.stepwise import StepwiseNetwork, default_lane_capacity
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m, rows = itscp_micro_tables(g)
    rng = np.random.default_rng(seed)
    t.set_micro_sources(np.concatenate([g["rand_draws"], rng.random(8 * len(g["rand_draws"]))]))     # (another schedule asks for more draws)
    rows = np.concatenate([rows] * 3)                                                                  # ... and may admit more vehicles
    gr, ptr = group_routes(rows, t.n_lanes)
    net = StepwiseNetwork(t, rows, cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=True)
    worst_q = worst_g = 0.0
    for k in range(3):
        act = rng.uniform(0.1, 0.9, len(g["action"])).astype(np.float32)
        o = _run(cuda, net, m, act)
        ref = oracle.net_hybrid(t, gr, ptr, act, *_args(m))
        assert ref["rc"] == 0 and o["counts"][0] == ref["n_spawned"] and o["counts"][3] == ref["draws_used"], k
        worst_q = max(worst_q, rel_max(o["queue"], ref["queue"]))
        assert abs(o["reward"] - ref["reward"]) <= 1e-5 * abs(ref["reward"])
        worst_g = max(worst_g, np.abs(o["grad"] - ref["g_action"]).max() / np.abs(ref["g_action"]).max())
    print("%s: persistent kernels vs oracle on random actions: queues %.1e, gradient %.1e" % (name, worst_q, worst_g))
    assert worst_q <= TOL_STATE and worst_g <= TOL_GRAD


@pytest.mark.parametrize("form", ["fused", "stepwise", "persistent"])
@pytest.mark.parametrize("name", ["micro_rv", "micro_rv_2x2", "micro_rv_l10", "hybrid_rv", "hybrid_rv_b", "hybrid_rv_d", "hybrid_rv_l10", "hybrid_rv_n2", "eval_hybrid_rv"])
def test_per_vehicle_idm_attributes_on_every_device_form(cuda, golden_dir, oracle, name, form):
    """Round 6 (dhts_hybrid_tables::veh_params): reference runs whose vehicles carry the attributes of a seeded
    MicroVehicle.random_micro_vehicle (micro_vehicle.py:75-121) through the fused kernels, the stepwise form and the persistent
    form: spawn count, queues <= 1e-5, reward, d reward / d action <= 1e-4 of the reference's run; the three forms agree with the oracle."""
    import torch
    from dhts import ops
    from dhts.network import group_routes
    from dhts.stepwise import StepwiseNetwork
    from test_oracle_golden import itscp_vehicle_params
    path = os.path.join(golden_dir, "itscp_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path)
    hard = name.startswith("eval")
    if "micro" in name:
        t, m, rows = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        rows = g["spawn_routes"]
    vp = itscp_vehicle_params(g)
    a = torch.tensor(g["action"], device=cuda, requires_grad=not hard)
    if form == "fused":
        try:
            t.check_kernel_limits()
        except ValueError:
            pytest.skip("beyond the fused kernels' one-workgroup limits")
        tab = ops.DeviceHybridTables(t, rows, cuda, vehicle_params=vp)
        if hard:
            reward, queue, counts = ops.net_hybrid_eval(a[None], tab, *_args(m))
            reward = reward[0]
        else:
            cut, reward, queue, counts = ops.net_hybrid_rollout(a[None], tab, *_args(m))
            cut.sum().backward()
            reward = reward[0]
        q, n_sp = queue[0].cpu().numpy(), int(counts[0, 0])
    else:
        net = StepwiseNetwork(t, rows, cuda, lane_capacity=32, persistent=form == "persistent", vehicle_params=vp)
        cut, reward, queue, counts = net.rollout(a, *_args(m), differentiable=not hard)
        if not hard:
            cut.backward()
        q, n_sp = queue.cpu().numpy(), int(counts[0])
    assert n_sp == m["n_vehicle_spawned"]
    assert state_report("%s (%s): queues vs reference" % (name, form), q.T, g["queue"]) <= TOL_STATE
    assert abs(float(reward.detach()) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    gr, ptr, gvp = group_routes(rows, t.n_lanes, vp)
    o = oracle.net_hybrid(t, gr, ptr, g["action"], *_args(m), vehicle_params=gvp, hard=hard)
    assert rel_max(q, o["queue"]) <= TOL_STATE
    if not hard:
        from test_oracle_golden import ILL_CONDITIONED_FULL_GRADIENT
        grad = a.grad.cpu().numpy()
        scale = np.abs(g["g_action"]).max()
        if name not in ILL_CONDITIONED_FULL_GRADIENT:
            assert np.abs(grad - g["g_action"]).max() <= TOL_GRAD * scale
        else:
            # (a fixture whose whole gradient moves by percents under a one-ulp change of the action,
            # tests/test_oracle_golden.py::test_itscp_network_with_per_vehicle_idm_attributes: held to the reference on the reward's first
            # t0 <= 360 of 480 steps)
            for t0, ref in zip(g["g_action_cut_steps"], g["g_action_cut"]):
                a2 = torch.tensor(g["action"], device=cuda, requires_grad=True)
                if form == "fused":
                    cut2, *_ = ops.net_hybrid_rollout(a2[None], tab, *_args(m), int(t0))
                    cut2.sum().backward()
                else:
                    cut2, *_ = net.rollout(a2, *_args(m), loss_steps=int(t0))
                    cut2.backward()
                assert np.abs(a2.grad.cpu().numpy() - ref).max() <= TOL_GRAD * scale, int(t0)
            # the whole-horizon gradient of this fixture moves by up to 8 % under a one-ulp change of the action (shown on the oracle in
            # the CPU test): two correct float32 implementations land on different sides of its knife edges.  A sanity bound only.
            assert np.isfinite(grad).all() and np.abs(grad - o["g_action"]).max() <= 0.1 * scale
            return
        assert np.abs(grad - o["g_action"]).max() <= TOL_GRAD * np.abs(o["g_action"]).max()


@pytest.mark.parametrize("name, seed", [("hybrid_p2", 11), ("hybrid_l10", 12), ("hybrid_n2", 13)])
def test_random_vehicle_attributes_and_actions_vs_oracle(cuda, golden_dir, oracle, name, seed):
    """Other signal schedules AND other vehicles: every route row of a reference network gets the attributes of a
    random_micro_vehicle(0.7 x speed limit) (rows are reused cyclically with their attributes), the action is random; fused, stepwise and
    persistent forms against the oracle: spawn counts, queues <= 1e-5, and the gradient of the reward's first half <= 1e-4."""
    import torch
    from dhts import ops
    from dhts.network import group_routes
    from dhts.stepwise import StepwiseNetwork
    from road.vehicle.micro_vehicle import MicroVehicle
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m = itscp_hybrid_tables(g)
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    rows = g["spawn_routes"][: max(2, len(g["spawn_routes"]) // 2)]            # fewer rows than vehicles: rows (and attributes) wrap around
    first = sorted(set(int(r[0]) for r in g["spawn_routes"]))
    have = set(int(r[0]) for r in rows)
    extra = [r for r in g["spawn_routes"] if int(r[0]) not in have]
    if extra:                                                                   # (every spawn lane needs at least one row)
        seen, keep = set(), []
        for r in extra:
            if int(r[0]) not in seen:
                seen.add(int(r[0]))
                keep.append(r)
        rows = np.concatenate([rows, np.asarray(keep)])
    assert set(int(r[0]) for r in rows) == set(first)
    vp = np.array([MicroVehicle.random_micro_vehicle(0.7 * m["speed_limit"]).params() for _ in rows])
    action = rng.uniform(0.1, 0.9, len(g["action"])).astype(np.float32)
    cut_t = m["T"] // 2
    gr, ptr, gvp = group_routes(rows, t.n_lanes, vp)
    o = oracle.net_hybrid(t, gr, ptr, action, *_args(m), vehicle_params=gvp, t_cut=cut_t)
    assert o["rc"] == 0 and o["n_spawned"] >= 2
    scale = np.abs(o["g_action"]).max()
    fits = True
    try:
        t.check_kernel_limits()
    except ValueError:
        fits = False
    for form in (["fused"] if fits else []) + ["stepwise", "persistent"]:
        a = torch.tensor(action, device=cuda, requires_grad=True)
        if form == "fused":
            cut, reward, queue, counts = ops.net_hybrid_rollout(a[None], ops.DeviceHybridTables(t, rows, cuda, vehicle_params=vp), *_args(m), cut_t)
            cut.sum().backward()
            q, n_sp = queue[0].cpu().numpy(), int(counts[0, 0])
        else:
            net = StepwiseNetwork(t, rows, cuda, lane_capacity=32, persistent=form == "persistent", vehicle_params=vp)
            cut, reward, queue, counts = net.rollout(a, *_args(m), loss_steps=cut_t)
            cut.backward()
            q, n_sp = queue.cpu().numpy(), int(counts[0])
        assert n_sp == o["n_spawned"], form
        assert state_report("%s random vehicles (%s): queues vs oracle" % (name, form), q, o["queue"]) <= TOL_STATE
        assert np.abs(a.grad.cpu().numpy() - o["g_action"]).max() <= TOL_GRAD * scale, form
