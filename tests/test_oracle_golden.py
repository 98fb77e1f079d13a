"""Pin the CPU oracle against the golden vectors generated from the imported reference
(tools/gen_goldens.py).  Runs without a GPU."""
import os

import numpy as np
import pytest

from util import grad_report, rel_elem, TOL_GRAD, TOL_STATE, meta_of, rel_max


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ---- G1/G2: interface known-answer vectors ----------------------------------------------------------------
def test_riemann_kat_bit_exact(oracle, golden_dir):
    g = load(golden_dir, "riemann_kat.npz")
    inp = g["inp"]
    combos = set()
    for i in range(len(inp)):
        L, R, um = inp[i, 0:4], inp[i, 4:8], inp[i, 8]
        case, q0, sp = oracle.arz_riemann(L, R, um)
        assert case == g["case"][i], i
        # double precision, same libm: bit-exact
        assert np.array_equal(q0, g["q0"][i]), (i, q0, g["q0"][i])
        assert np.array_equal(sp, g["speed"][i]), i
        dL, dR = oracle.arz_dLdR(case, q0, L, R, um)
        assert np.array_equal(dL, g["dL"][i]) and np.array_equal(dR, g["dR"][i]), i
        assert np.array_equal(oracle.arz_flux_prime(q0, um), g["fp"][i]), i
        combos.add((int(g["branch"][i]), case))
    # all 11 (branch, case) pairs of the solver are covered
    assert combos == {(1, 0), (2, 0), (2, 2), (3, 0), (4, 0), (4, 1), (5, 0), (5, 1), (5, 2), (6, 0), (6, 2)}


# ---- G3: one dMacroLane step --------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["rand64", "sanity100", "vacuum9", "single1", "jam33"])
def test_macro_step(oracle, golden_dir, name):
    g = load(golden_dir, "macro_step.npz")
    c = meta_of(g)["configs"][name]
    o = oracle.macro_step(g[name + "_state"], c["dt"], c["dx"], c["u_max"])
    assert o["rc"] == 0
    assert np.array_equal(o["case"], g[name + "_case"])
    assert np.array_equal(o["speed"], g[name + "_speed"])
    assert np.array_equal(o["nr"], g[name + "_nr"])
    assert np.array_equal(o["ny"], g[name + "_ny"])
    assert np.array_equal(o["dqs"], g[name + "_dqs"])          # Jacobian tape: bit-exact
    # u / u_eq glue: torch's float32 sqrt is not correctly rounded on ~0.6 % of inputs -> 1 ulp tolerance
    assert rel_max(o["nu"], g[name + "_nu"]) <= 2e-7
    assert rel_max(o["nueq"], g[name + "_nueq"]) <= 2e-7
    assert np.mean(o["nu"] != g[name + "_nu"]) <= 0.05
    g_r, g_y = oracle.macro_step_bwd(g[name + "_dqs"], g[name + "_g_nr"], g[name + "_g_ny"])
    assert np.array_equal(g_r, g[name + "_g_r"]) and np.array_equal(g_y, g[name + "_g_y"])


# ---- G4: macro rollouts ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["small", "c1", "sanity", "bench64", "long", "c2slice", "x3", "x5", "x10", "x22", "x31", "x40"])
def test_macro_rollout(oracle, golden_dir, name):
    g = load(golden_dir, "macro_rollout_%s.npz" % name)
    m = meta_of(g)
    f = oracle.macro_rollout_fwd(g["r0"], g["u0"], g["ghost_r"], g["ghost_u"], m["T"], m["dt"], m["dx"], m["u_max"],
                                 want_hist=True)
    assert f["rc"] == 0
    assert rel_max(f["rT"][0], g["rT"]) <= TOL_STATE
    assert rel_max(f["yT"][0], g["yT"]) <= TOL_STATE
    assert rel_max(f["uT"][0], g["uT"]) <= TOL_STATE
    # element-wise too (north_star: "state <= 1e-5 relative"), with a floor of 1e-6 max|ref| under the division
    e_elem = max(rel_elem(f["rT"][0], g["rT"]), rel_elem(f["uT"][0], g["uT"]))
    print("G4 %s: element-wise state error %.2e" % (name, e_elem))
    assert e_elem <= TOL_STATE
    for t in range(len(g["steps_r"])):
        assert rel_max(f["hist_r"][t, 0], g["steps_r"][t]) <= TOL_STATE
        assert rel_max(f["hist_u"][t, 0], g["steps_u"][t]) <= TOL_STATE
        assert rel_elem(f["hist_r"][t, 0], g["steps_r"][t]) <= TOL_STATE and rel_elem(f["hist_u"][t, 0], g["steps_u"][t]) <= TOL_STATE
    if m["tap"] == "final_sq":
        b = oracle.macro_rollout_bwd(f, g_rT=2 * f["rT"], g_uT=2 * f["uT"])
        loss = float(np.sum(f["rT"].astype(np.float64) ** 2) + np.sum(f["uT"].astype(np.float64) ** 2))
    else:
        ones = np.ones_like(f["hist_r"])
        b = oracle.macro_rollout_bwd(f, gh_r=ones, gh_y=ones, gh_u=ones)
        loss = float(f["hist_r"].sum(dtype=np.float64) + f["hist_y"].sum(dtype=np.float64) + f["hist_u"].sum(dtype=np.float64))
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert grad_report("G4 %s d loss / d r0" % name, b["g_r0"][0], g["g_r0"]) <= TOL_GRAD
    assert grad_report("G4 %s d loss / d u0" % name, b["g_u0"][0], g["g_u0"]) <= TOL_GRAD
    assert rel_max(b["g_ghost_r"][0], g["g_ghost_r"]) <= TOL_GRAD
    assert rel_max(b["g_ghost_u"][0], g["g_ghost_u"]) <= TOL_GRAD


# ---- G5: IDM known-answer vectors -----------------------------------------------------------------------------
def test_idm_kat_bit_exact(oracle, golden_dir):
    g = load(golden_dir, "idm_kat.npz")
    flags_seen = set()
    for i, (a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt) in enumerate(g["inp"]):
        acc, s, fl = oracle.idm_acc(a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt)
        assert acc == g["acc"][i] and s == g["sstar"][i] and fl == tuple(g["flags"][i]), i
        dE, dLd = oracle.idm_jac(a_max, a_pref, v, v_t, dp, dv, s0, Tp, s, dt, fl)
        assert np.array_equal(dE, g["dEgo"][i]) and np.array_equal(dLd, g["dLeading"][i]), i
        flags_seen.add(fl)
    assert {(0, 0), (1, 0), (0, 1)} <= flags_seen     # both clips exercised
    # G5b: raw gaps below 1e-5 -- acceleration, spacing and flags from the clamped gap, Jacobians from the raw one (dmicro_lane.py:97)
    g = load(golden_dir, "idm_kat_smallgap.npz")
    for i, (a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt) in enumerate(g["inp"]):
        acc, s, fl = oracle.idm_acc(a_max, a_pref, v, v_t, max(dp, 1e-5), dv, s0, Tp, dt)
        assert acc == g["acc"][i] and s == g["sstar"][i] and fl == tuple(g["flags"][i]), i
        dE, dLd = oracle.idm_jac(a_max, a_pref, v, v_t, dp, dv, s0, Tp, s, dt, fl)
        assert np.array_equal(dE, g["dEgo"][i]) and np.array_equal(dLd, g["dLeading"][i]), i


# ---- G6: micro rollouts: bit-exact state AND gradients ---------------------------------------------------------
@pytest.mark.parametrize("name", ["inv10", "rand24", "dense16", "long", "c3slice", "x2", "x9", "x15", "x23", "x37", "x44"])
def test_micro_rollout_bit_exact(oracle, golden_dir, name):
    g = load(golden_dir, "micro_rollout_%s.npz" % name)
    m = meta_of(g)
    f = oracle.micro_rollout_fwd(g["p0"], g["v0"], g["params"], m["T"], m["dt"], m["head"][0], m["head"][1], want_hist=True)
    assert f["rc"] == 0
    assert np.array_equal(f["pT"][0], g["pT"]) and np.array_equal(f["vT"][0], g["vT"])
    for t in range(len(g["steps_p"])):
        assert np.array_equal(f["hist_p"][t, 0], g["steps_p"][t])
        assert np.array_equal(f["hist_v"][t, 0], g["steps_v"][t])
    if m["tap"] == "final_sq":
        b = oracle.micro_rollout_bwd(f, g_pT=np.float32(2e-4) * f["pT"], g_vT=2 * f["vT"])
    else:
        ones = np.ones_like(f["hist_p"])
        b = oracle.micro_rollout_bwd(f, gh_p=ones, gh_v=ones)
    assert np.array_equal(b["g_p0"][0], g["g_p0"]) and np.array_equal(b["g_v0"][0], g["g_v0"])


# ---- error conventions ----------------------------------------------------------------------------------------
def test_cfl_violation_reported(oracle):
    # dt * u_max / dx > 1: the reference asserts (_macro_lane.py:141-146)
    st = np.zeros((4, 6), np.float32)
    st[0] = 0.3
    st[2] = 20.0
    for i in range(6):
        st[1, i], st[3, i] = oracle.arz_from_r_u(st[0, i], st[2, i], 30.0)
    assert oracle.macro_step(st, 1.0, 5.0, 30.0)["rc"] == oracle.ERR_CFL
    assert oracle.macro_step(st, 0.01, 5.0, 30.0)["rc"] == 0


def test_collision_reported(oracle):
    par = np.tile(np.array([30.0, 24.0, 27.0, 0.5, 0.1, 5.0]), (2, 1))
    o = oracle.micro_step(np.array([0.0, 3.0], np.float32), np.array([10.0, 1.0], np.float32), par, 1000.0, 0.0, 0.01)
    assert o["rc"] == oracle.ERR_COLLISION and o["err_index"] == 0
    assert np.all(np.isfinite(o["np"])) and np.all(np.isfinite(o["nv"]))


# ---- G8: macro road network with signals (itscp `macro` mode) ---------------------------------------------------------
def itscp_tables(g):
    from dhts.network import SIG_ALWAYS, SIG_NS, SIG_WE, MacroNetworkTables
    m = meta_of(g)
    tab = g["lane_tab"]
    kinds = []
    for s in g["lane_str"]:
        loc, _, app = str(s).split("|")
        kinds.append(SIG_ALWAYS if (loc == "mid" or app == "0") else (SIG_WE if loc in ("west", "east") else SIG_NS))
    inter = (tab[:, 5] * m["num_intersection"] + tab[:, 6]).astype(int)
    return MacroNetworkTables(tab[:, 3].astype(int), tab[:, 2], g["edges"], kinds, inter, g["macro_route"], g["schedule"]), m


@pytest.mark.parametrize("name", ["macro_small", "macro", "macro_2x2", "macro_half", "macro_long", "macro_3x3x3", "sweep_b"])      # sweep_*: ref_sweep.py's shapes
def test_itscp_macro_network(oracle, golden_dir, name):
    g = load(golden_dir, "itscp_%s.npz" % name)
    t, m = itscp_tables(g)
    o = oracle.net_macro(t, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                         1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
    assert o["rc"] == 0
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert rel_elem(o["queue"].T, g["queue"]) <= 10 * TOL_STATE        # a queue term is (sum of sigmoids)^2 dt: twice the state's relative error and the sigmoids' slope
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert grad_report("G8 %s d reward / d action" % name, o["g_action"], g["g_action"]) <= TOL_GRAD


@pytest.mark.parametrize("name", ["eval_macro", "eval_macro_2x2", "eval_macro_3x3x3", "sweep_a"])
def test_itscp_macro_network_evaluation_episode(oracle, golden_dir, name):
    """ItscpEnv.step(action, False) of the reference (what Trainer.evaluate runs): hard signals, hard ghost switch, hard
    is_static.  The queue terms are squares of sums of whole cells' vehicle counts, so a cell whose speed crosses
    static_speed one step early or late would show as a large error: equality of every term is the test."""
    g = load(golden_dir, "itscp_%s.npz" % name)
    t, m = itscp_tables(g)
    assert m["differentiable"] is False
    o = oracle.net_macro(t, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                         1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"], hard=True)
    assert o["rc"] == 0
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert rel_elem(o["queue"].T, g["queue"]) <= 10 * TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    # and the differentiable episode of the same inputs is a different number (the switch does something)
    o2 = oracle.net_macro(t, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                          1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"], want_grad=False)
    assert abs(o2["reward"] - o["reward"]) > 1e-3 * abs(o["reward"])


def itscp_hybrid_tables(g):
    from dhts.network import SIG_ALWAYS, SIG_NS, SIG_WE, HybridNetworkTables
    m = meta_of(g)
    tab = g["lane_tab"]
    kinds = []
    for s in g["lane_str"]:
        loc, _, app = str(s).split("|")
        kinds.append(SIG_ALWAYS if (loc == "mid" or app == "0") else (SIG_WE if loc in ("west", "east") else SIG_NS))
    inter = (tab[:, 5] * m["num_intersection"] + tab[:, 6]).astype(int)
    return HybridNetworkTables(tab[:, 1].astype(int), tab[:, 3].astype(int), tab[:, 2], g["edges"], kinds, inter,
                               g["macro_route"], g["schedule"]), m


FULL_HORIZON_600 = ["hybrid_half", "hybrid_s2", "hybrid_s3", "hybrid_p2_600"]     # run_itscp_hybrid.sh's episode, 600 steps


# round 5: hybrid_l30 (30 m lanes), hybrid_5x5 (144 IDM lanes) and hybrid_n2l30 (252 lanes + 1 300 cells) pin the oracle where the
# fused kernels do not reach (or only just): the stepwise device path is judged by it there
@pytest.mark.parametrize("name", ["hybrid_short", "hybrid_p2", "hybrid_p3", "hybrid_l10", "hybrid_n2", "hybrid_4x4", "hybrid", "hybrid_l30",
                                  "hybrid_5x5", "hybrid_n2l30", "sweep_g"] + FULL_HORIZON_600)
def test_itscp_hybrid_network(oracle, golden_dir, name):
    """G8 hybrid: macro lanes, micro lanes, spawns, lane changes and deposits against the reference's own run.
    FULL_HORIZON_600 = four reference runs of BASELINE config 4's exact episode (3 x 3 intersections, 1 lane, 5 m, 20 s,
    signal 4 s: 600 steps, 45 actions) -- action 0.5 everywhere and three random actions over two inflow patterns: the
    WHOLE d reward / d action must match to 1e-4 (achieved: <= 2e-6), and so must the gradient of the reward restricted to
    its first 150 / 300 / 450 / 540 steps."""
    if not os.path.exists(os.path.join(golden_dir, "itscp_%s.npz" % name)):
        pytest.skip("golden not generated")
    g = load(golden_dir, "itscp_%s.npz" % name)
    t, m = itscp_hybrid_tables(g)
    from dhts.network import group_routes
    routes, route_ptr = group_routes(g["spawn_routes"], t.n_lanes)
    run = lambda **kw: oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2,     # noqa: E731
                                         m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"],
                                         m["speed_limit"], m["static_speed"], m["vehicle_length"], **kw)
    o = run()
    assert o["rc"] == 0 and o["n_spawned"] == m["n_vehicle_spawned"]
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    scale = np.abs(g["g_action"]).max()
    assert np.abs(o["g_action"] - g["g_action"]).max() <= TOL_GRAD * scale
    if name in FULL_HORIZON_600 or name == "hybrid":
        assert m["T"] == 600 and len(g["action"]) == 45
        assert np.abs(o["g_action"] - g["g_action"]).max() <= 1e-5 * scale          # achieved 2e-6 (hybrid: 3.5e-6)
    if name in FULL_HORIZON_600:
        for t0, ref in zip(g["g_action_cut_steps"], g["g_action_cut"]):
            oc = run(t_cut=int(t0))
            assert np.abs(oc["g_action"] - ref).max() <= 1e-5 * scale, int(t0)
    if name == "hybrid":
        # the first 600-step golden: from step 480 lane 16 holds a standing vehicle whose late loss terms reach the action through
        # ~90 steps of 1.144-fold amplification; the gradient of the reward RESTRICTED to its first t0 steps is pinned up to 540
        # (achieved: 3.3e-6 for t0 <= 510, 7.8e-5 at t0 = 540; t0 = 570: test_restricted_gradient_lattice_of_the_standing_vehicle)
        assert o["n_deposits"] == 12
        for t0, ref in zip(g["g_action_cut_steps"], g["g_action_cut"]):
            if t0 <= 540:
                oc = run(t_cut=int(t0))
                assert np.abs(oc["g_action"] - ref).max() <= (1e-5 if t0 <= 510 else TOL_GRAD) * scale, int(t0)


def test_restricted_gradient_lattice_of_the_standing_vehicle(oracle, golden_dir):
    """Why `itscp_hybrid.npz`'s g_action_cut at t0 = 570 is the one reference number no test holds to 1e-4 (it differs by 7.5e-3
    max|g| from oracle and kernels, which agree with each other to 1.2e-7 there, tests/test_hybrid_gpu.py).  Lane 16's loss
    terms of steps >= 540 reach the action through the standing vehicle's ~90 steps of 1.144-fold amplification: a float32 ulp of
    the cotangent where the terms join becomes a LATTICE of 2.2e-2 = 1.25e-3 max|g| in the action gradient.  Shown here on the
    restatement alone: (1) the whole gradient matches the reference (3.5e-6: all terms join before the amplification);
    (2) the increments of the restricted gradient from one t0 to the next are whole multiples of one quantum q >= 1e-3 max|g|;
    (3) the reference's own t0 = 570 number sits on the same lattice, a whole number of quanta (six) away -- which terms are
    summed first in float32 (autograd's engine order in the reference, the reverse sweep's order here) decides the lattice
    point, and no restatement short of torch's engine reproduces that choice."""
    g = load(golden_dir, "itscp_hybrid.npz")
    t, m = itscp_hybrid_tables(g)
    from dhts.network import group_routes
    routes, route_ptr = group_routes(g["spawn_routes"], t.n_lanes)
    run = lambda **kw: oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2,     # noqa: E731
                                         m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"],
                                         m["speed_limit"], m["static_speed"], m["vehicle_length"], **kw)
    scale = np.abs(g["g_action"]).max()
    assert np.abs(run()["g_action"] - g["g_action"]).max() <= 1e-5 * scale                      # (1)
    ref570 = g["g_action_cut"][list(g["g_action_cut_steps"]).index(570)]
    cuts = {tc: run(t_cut=tc)["g_action"] for tc in range(562, 579)}
    d570 = cuts[570] - ref570
    k = int(np.argmax(np.abs(d570)))
    assert 1e-3 * scale < abs(d570[k]) < 1e-2 * scale                                            # the mismatch no test holds to 1e-4
    inc = np.array([cuts[tc + 1][k] - cuts[tc][k] for tc in range(562, 578)], dtype=np.float64)
    q = None
    for n in range(1, 33):                                                                       # the lattice constant: the coarsest q = min|inc| / n
        cand = np.abs(inc).min() / n
        if np.abs(inc / cand - np.round(inc / cand)).max() <= 0.02:
            q = cand
            break
    assert q is not None and q >= 1e-3 * scale, (q, inc)                                         # (2) quantum 2.2e-2 = 1.25e-3 max|g|
    n_q = d570[k] / q
    assert abs(n_q - round(n_q)) <= 0.05 and 1 <= abs(round(n_q)) <= 8, n_q                      # (3) six quanta
    # every other component of the mismatch is a few quanta of its own (smaller) lattice: nothing above the largest one
    assert np.abs(d570).max() <= 8 * q


@pytest.mark.parametrize("name", ["eval_hybrid_short", "eval_hybrid_p2", "eval_hybrid", "eval_hybrid_4x4", "eval_hybrid_n2l30", "eval_hybrid_5x5", "sweep_f"])
def test_itscp_hybrid_network_evaluation_episode(oracle, golden_dir, name):
    """Evaluation episodes of the hybrid network (240 steps; 480 steps over problem_2's inflows; BASELINE config 4's 600-step
    episode): hard signals and boundaries, head gap = green iff the lane's own signal >= 0.5, hard is_static for cells and
    vehicles; same spawns and deposits as the reference's run."""
    g = load(golden_dir, "itscp_%s.npz" % name)
    t, m = itscp_hybrid_tables(g)
    from dhts.network import group_routes
    routes, route_ptr = group_routes(g["spawn_routes"], t.n_lanes)
    o = oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                          1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"], hard=True)
    assert o["rc"] == 0 and o["n_spawned"] == m["n_vehicle_spawned"]
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))


def itscp_micro_tables(g):
    """itscp `micro` mode fixture -> tables: every lane an IDM lane; the waiting routes of every lane in admission order (the
    reference pops its waiting list from the end) as the route rows; the recorded admission draws."""
    import json
    t, m = itscp_hybrid_tables(g)
    assert t.n_cells == 0 and t.lane_source.sum() > 0
    waiting = {int(l): r for l, r in json.loads(str(g["waiting_routes"])).items()}
    rows = []
    for l in range(t.n_lanes):
        for r in reversed(waiting.get(l, [])):
            rows.append(list(r) + [-1] * (32 - len(r)))
    t.set_micro_sources(g["rand_draws"])
    return t, m, np.asarray(rows, dtype=np.int32)


def itscp_vehicle_params(g):
    """[n_routes][6] IDM attributes in the order of the fixture's route rows (spawn_routes for a hybrid network; the waiting lists in
    admission order, as itscp_micro_tables lays them out, for `micro` mode), or None for a fixture of default vehicles."""
    import json
    if "veh_params" not in g.files:
        return None
    if g["veh_params"].shape[0]:
        return np.ascontiguousarray(g["veh_params"], dtype=np.float64)
    waiting = {int(l): p for l, p in json.loads(str(g["waiting_params"])).items()}
    if not any(len(p) for p in waiting.values()):
        return None
    routes = {int(l): r for l, r in json.loads(str(g["waiting_routes"])).items()}
    rows = []
    for l in sorted(routes):
        assert len(routes[l]) == len(waiting.get(l, []))
        for p in reversed(waiting.get(l, [])):
            rows.append(p)
    return np.asarray(rows, dtype=np.float64)


@pytest.mark.parametrize("name", ["micro_small", "micro", "micro_2x2", "micro_p2", "micro_l10", "micro_jam_a", "micro_jam_b", "micro_jam_c", "sweep_c", "sweep_e"])
def test_itscp_micro_mode_network(oracle, golden_dir, name):
    """itscp `micro` mode (run_itscp_micro.sh: 40 IDM lanes, 65 vehicles admitted stochastically over 300 steps; and a 16-lane
    case): source lanes admit waiting vehicles against the host's recorded draws (_simulator.py:153-174), every recorded draw
    is consumed, same vehicle count, queues, reward and d reward / d action as the reference's run.  (The reference steps
    these lanes with the plain autodiff MicroLane in float32 TENSOR arithmetic; since round 5 the restatement follows that ladder
    operation by operation in this mode (oracle_micro_step_f32): queues 6e-8 / 2.2e-6 / 1.7e-7 on the three goldens, where the
    analytic operator's float64 ladder gave 1.2e-5 / 3.7e-6 / 2.7e-6.)
    micro_jam_*: congested 8-second episodes (~110 vehicles on 60 m lanes) whose followers close in below POSITION_DELTA_EPS and collide:
    autograd differentiates the forward's clamps there (constants), the gradient is finite -- dIDM's formulas at the un-clamped gap
    (the hybrid lanes' rule) divide by zero on all three."""
    g = load(golden_dir, "itscp_%s.npz" % name)
    t, m, rows = itscp_micro_tables(g)
    from dhts.network import group_routes
    routes, route_ptr = group_routes(rows, t.n_lanes)
    o = oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                          1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
    assert o["rc"] == 0 and o["n_spawned"] == m["n_vehicle_spawned"] and o["draws_used"] == len(g["rand_draws"])
    assert rel_max(o["queue"].T, g["queue"]) <= 1e-5
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert grad_report("G8 %s d reward / d action" % name, o["g_action"], g["g_action"]) <= TOL_GRAD



@pytest.mark.parametrize("name, cap", [("hybrid_n2l30", 8), ("hybrid_l30", 8), ("hybrid_5x5", 8), ("hybrid_p2", 8), ("macro_3x3x3", 4)])
def test_default_lane_capacity_follows_the_geometry(golden_dir, name, cap):
    """dhts.stepwise.default_lane_capacity: the longest micro lane in vehicles + 2 as a power of two in 4 .. 32 (host logic)."""
    from dhts.stepwise import default_lane_capacity
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m = itscp_hybrid_tables(g) if name.startswith("hybrid") else itscp_tables(g)
    assert default_lane_capacity(t, m["vehicle_length"]) == cap
    assert default_lane_capacity(t, m["vehicle_length"] / 100.0) == (32 if name.startswith("hybrid") else 4)      # (the ceiling)


def test_persistent_form_pays_up_to_a_workgroup_of_lanes(golden_dir):
    from dhts.stepwise import persistent_form_pays
    g = np.load(os.path.join(golden_dir, "itscp_hybrid_5x5.npz"))
    t, _ = itscp_hybrid_tables(g)
    assert persistent_form_pays(t)
    import copy
    big = copy.copy(t)
    big.n_lanes = 1296
    assert not persistent_form_pays(big)


@pytest.mark.parametrize("name", ["eval_micro_small", "eval_micro", "eval_micro_2x2", "sweep_d"])
def test_itscp_micro_mode_evaluation_episode(oracle, golden_dir, name):
    """Evaluation episodes in `micro` mode (round 5 fixtures): without gradients nothing in the reference is a tensor -- Python floats all
    the way --, so the lanes step in the analytic operator's float64 ladder here (not the float32 tensor ladder of the differentiable
    episodes); hard signals, hard is_static; every recorded admission draw consumed, same vehicles, queues and reward."""
    g = load(golden_dir, "itscp_%s.npz" % name)
    t, m, rows = itscp_micro_tables(g)
    from dhts.network import group_routes
    routes, route_ptr = group_routes(rows, t.n_lanes)
    o = oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                          1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"], hard=True)
    assert o["rc"] == 0 and o["n_spawned"] == m["n_vehicle_spawned"] and o["draws_used"] == len(g["rand_draws"])
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))


def test_running_mean_as_numpy_computes_it(oracle, golden_dir):
    """Where the 2e-6 of the `micro`-mode fixtures come from: the reference keeps its RunningMean samples in a float32 array and takes
    np.mean of it at every sample (example/common/rms.py:8-22) -- numpy's PAIRWISE float32 summation (eight accumulators per block of up
    to 128 elements, blocks halved above that, the ufunc's 8192-element buffer at a time), O(window) per sample.  With the oracle's means
    evaluated that way EVERY queue term of `micro_small` and of `micro` is the reference's bit for bit (gradient 2e-7); the default, and
    the kernels, keep the exact float64 prefix mean that summation approximates."""
    from dhts.network import group_routes
    res = {}
    try:
        for mode in (0, 1):
            oracle.set_numpy_mean(mode)
            for name in ("micro_small", "micro"):
                g = load(golden_dir, "itscp_%s.npz" % name)
                t, m, rows = itscp_micro_tables(g)
                routes, route_ptr = group_routes(rows, t.n_lanes)
                o = oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                                      1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
                res[(mode, name)] = (rel_max(o["queue"].T, g["queue"]), float(np.abs(o["g_action"] - g["g_action"]).max() / np.abs(g["g_action"]).max()))
    finally:
        oracle.set_numpy_mean(0)
    print(res)
    assert res[(1, "micro")][0] == 0.0 and res[(1, "micro")][1] <= 5e-7 and res[(0, "micro")][0] > 1e-6
    assert res[(1, "micro_small")][0] == 0.0 and res[(1, "micro_small")][1] <= 2e-7


def test_glue_square_root_as_torch_computes_it(oracle, golden_dir):
    """Where the last 1e-6 of the macro and hybrid fixtures come from: u_eq of a float32 tensor is (r + eps) ** 0.5 = torch's CPU float32
    square root, and on the build the goldens were generated with (this container's) that kernel is not correctly rounded -- one ulp low
    for 0.6 % of the arguments -- while sqrtf, and the device's, is.  With torch.sqrt handed to the oracle's glue AND the running
    means as numpy computes them, EVERY queue term of the macro and hybrid networks is the reference's bit for bit (12 000 of them on
    `macro`, 34 560 on `hybrid_short` -- whose head vehicles step in the reference's mixed float32 / double arithmetic,
    oracle_micro_head_mixed); neither is a property of the reference's algorithm, so the oracle's defaults and the kernels keep IEEE's
    square root and the exact mean."""
    import torch
    from dhts.network import group_routes
    probe = torch.full((), 0.16979104280471802, dtype=torch.float32)
    if float(torch.sqrt(probe)) != float(np.float32(0.41205707)):
        pytest.skip("this torch build rounds its float32 sqrt correctly: not the environment the goldens were generated in")
    buf = torch.zeros((), dtype=torch.float32)

    def torch_sqrt(x):
        buf.fill_(x)
        return torch.sqrt(buf).item()

    res = {}
    try:
        for name in ("macro_small", "macro", "hybrid_short"):
            g = load(golden_dir, "itscp_%s.npz" % name)
            t, m = itscp_hybrid_tables(g)
            rows = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
            routes, route_ptr = group_routes(rows, t.n_lanes)
            for mode in (0, 1):
                oracle.set_numpy_mean(mode)
                oracle.set_sqrtf_hook(torch_sqrt if mode else None)
                o = oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                                      1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
                assert o["rc"] == 0
                res[(mode, name)] = rel_max(o["queue"].T, g["queue"])
    finally:
        oracle.set_numpy_mean(0)
        oracle.set_sqrtf_hook(None)
    print(res)
    assert res[(1, "macro_small")] == 0.0 < res[(0, "macro_small")]
    assert res[(1, "macro")] == 0.0 < res[(0, "macro")] and res[(1, "hybrid_short")] == 0.0 < res[(0, "hybrid_short")]
    # the straight lanes (no running mean there): with torch's square root the oracle's final state is the reference's BIT FOR BIT on
    # every G4 rollout -- BASELINE config 1 (100 cells x 200 steps) and one lane of config 2 (512 cells x 1000 steps) among them
    differing = {}
    try:
        for name in ("small", "sanity", "c1", "bench64", "long", "c2slice"):
            g = load(golden_dir, "macro_rollout_%s.npz" % name)
            m = meta_of(g)
            for mode in (0, 1):
                oracle.set_sqrtf_hook(torch_sqrt if mode else None)
                f = oracle.macro_rollout_fwd(g["r0"], g["u0"], g["ghost_r"], g["ghost_u"], m["T"], m["dt"], m["dx"], m["u_max"])
                assert f["rc"] == 0
                differing[(mode, name)] = int((f["rT"][0] != g["rT"]).sum() + (f["yT"][0] != g["yT"]).sum() + (f["uT"][0] != g["uT"]).sum())
    finally:
        oracle.set_sqrtf_hook(None)
    print(differing)
    assert all(differing[(1, name)] == 0 for name in ("small", "sanity", "c1", "bench64", "long", "c2slice"))
    assert differing[(0, "c2slice")] > 500 and differing[(0, "c1")] > 0


def test_source_ghost_in_double(oracle, golden_dir):
    """The reference keeps the upstream ghost of a SOURCE lane as Python floats (_simulator.py:68-71: inflow from the schedule,
    u = u_eq(r) in double) and its Riemann solve reads them as such; until the end of round 5 oracle and kernels rounded them to float32
    like every other ghost cell, which was the first half of the hybrid fixtures' 1.5-4e-6 (tools/probes/ref_state_trace.py: the macro
    cells are bit-identical to the reference's for the first 22 steps with the doubles, and differ from step 0 without).  The double
    ghost is the default now (oracle_set_source_ghost_f64(0) restores the rounding): the queue terms are closer on every hybrid fixture
    tried."""
    from dhts.network import group_routes
    res = {}
    try:
        for mode in (0, 1):
            oracle.set_source_ghost_f64(mode)
            for name in ("hybrid_short", "hybrid_p2", "hybrid_n2"):
                g = load(golden_dir, "itscp_%s.npz" % name)
                t, m = itscp_hybrid_tables(g)
                routes, route_ptr = group_routes(g["spawn_routes"], t.n_lanes)
                o = oracle.net_hybrid(t, routes, route_ptr, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                                      1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
                assert o["rc"] == 0 and o["n_spawned"] == m["n_vehicle_spawned"]
                res[(mode, name)] = rel_max(o["queue"].T, g["queue"])
    finally:
        oracle.set_source_ghost_f64(1)
    print(res)
    for name in ("hybrid_short", "hybrid_p2", "hybrid_n2"):
        assert res[(1, name)] < res[(0, name)] <= TOL_STATE
    assert res[(1, "hybrid_p2")] <= 0.7 * res[(0, "hybrid_p2")]


ILL_CONDITIONED_FULL_GRADIENT = {"hybrid_rv"}


@pytest.mark.parametrize("name", ["micro_rv", "micro_rv_2x2", "micro_rv_l10", "hybrid_rv", "hybrid_rv_b", "hybrid_rv_d", "hybrid_rv_l10", "hybrid_rv_n2", "eval_hybrid_rv"])
def test_itscp_network_with_per_vehicle_idm_attributes(oracle, golden_dir, name):
    """Round 6: vehicles that are NOT default_micro_vehicle -- reference runs whose vehicles take the attributes of a seeded
    MicroVehicle.random_micro_vehicle (road/vehicle/micro_vehicle.py:75-121; tools/gen_goldens.py random_vehicles): the restatement with
    the per-vehicle table beside the routes (dhts_hybrid_tables::veh_params) reproduces spawn counts, queues, reward and
    d reward / d action; with default vehicles it does not."""
    if not os.path.exists(os.path.join(golden_dir, "itscp_%s.npz" % name)):
        pytest.skip("golden not generated")
    from dhts.network import group_routes
    g = load(golden_dir, "itscp_%s.npz" % name)
    hard = name.startswith("eval")
    if "micro" in name:
        t, m, rows = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        rows = g["spawn_routes"]
    vp = itscp_vehicle_params(g)
    assert vp is not None and vp.shape == (rows.shape[0], 6) and not np.allclose(vp[:, 0], vp[0, 0])
    routes, route_ptr, gvp = group_routes(rows, t.n_lanes, vp)
    args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])
    o = oracle.net_hybrid(t, routes, route_ptr, g["action"], *args, vehicle_params=gvp, hard=hard)
    assert o["rc"] == 0 and o["n_spawned"] == m["n_vehicle_spawned"] and m["n_vehicle_spawned"] >= 4
    assert rel_max(o["queue"].T, g["queue"]) <= TOL_STATE
    assert abs(o["reward"] - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    run = lambda a, **kw: oracle.net_hybrid(t, routes, route_ptr, a, *args, vehicle_params=gvp, **kw)      # noqa: E731
    scale = np.abs(g["g_action"]).max() if not hard else 1.0
    if not hard and name not in ILL_CONDITIONED_FULL_GRADIENT:
        assert grad_report("G8 %s d reward / d action" % name, o["g_action"], g["g_action"]) <= TOL_GRAD
    if not hard and "g_action_cut_steps" in g.files:
        for t0, ref in zip(g["g_action_cut_steps"], g["g_action_cut"]):
            assert np.abs(run(g["action"], t_cut=int(t0))["g_action"] - ref).max() <= TOL_GRAD * scale, int(t0)
    if name in ILL_CONDITIONED_FULL_GRADIENT:
        # 10 deposits, the last ones standing behind a red light for the rest of the 480 steps (the 1.144-per-step amplification of
        # test_restricted_gradient_lattice_of_the_standing_vehicle, 685 equal-speed decisions within 5e-8 of the solver's threshold):
        # the gradient of the reward's first 120 / 240 / 360 steps is the reference's to 1e-7 (above), but the WHOLE gradient moves by
        # percents when the action moves by one float32 ulp -- no restatement can be held to 1e-4 of a number like that, and the
        # kernels are held to the restatement instead (tests/test_stepwise_gpu.py).  Shown here:
        rng = np.random.default_rng(0)
        moved = []
        for _ in range(3):
            a = g["action"].copy()
            for i in range(len(a)):
                if rng.random() < 0.5:
                    a[i] = np.nextafter(a[i], np.float32(2.0) if rng.random() < 0.5 else np.float32(-2.0))
            moved.append(np.abs(run(a)["g_action"] - o["g_action"]).max() / scale)
        assert max(moved) > 10 * TOL_GRAD, moved
        assert np.abs(o["g_action"] - g["g_action"]).max() <= max(moved) * scale          # ... and the reference's number is within that spread
    d = oracle.net_hybrid(t, routes, route_ptr, g["action"], *args, hard=hard)
    assert d["rc"] != 0 or rel_max(d["queue"].T, g["queue"]) > 1e-3            # the attributes matter
