"""The reference's operator surface (road.lane / road.network / model.* import paths) on top of the HIP kernels,
driven the way the reference's own examples and sanity scripts drive it (example/inverse/macro.py:34-125,
example/inverse/micro.py:36-118, example/sanity/macro.py:45-129) and compared with the golden vectors."""
import os

import numpy as np
import pytest

from util import TOL_GRAD, TOL_STATE, meta_of, rel_max, ulp_diff

pytestmark = pytest.mark.gpu


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def tt(x, dev, grad=False):
    import torch
    t = torch.tensor(np.ascontiguousarray(x), device=dev)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("name", ["rand64", "vacuum9", "single1", "jam33"])
def test_dmacro_lane_step_like_reference(cuda, golden_dir, name):
    """lane.forward(dt) -> next state, lane.d_lane[-1].dqs, and the operator's backward (G3)."""
    import torch
    from road.lane.dmacro_lane import dMacroForwardLayer, dMacroLane
    g = load(golden_dir, "macro_step.npz")
    c = meta_of(g)["configs"][name]
    st = g[name + "_state"]
    lane = dMacroLane(0, c["N"] * c["dx"], c["u_max"], c["dx"])
    assert lane.num_cell == c["N"] and abs(lane.cell_length - c["dx"]) < 1e-12
    lane.set_state_vector_u(tt(st[0, 1:-1], cuda), tt(st[2, 1:-1], cuda))
    lane.set_leftmost_cell(tt(st[0, 0], cuda), tt(st[2, 0], cuda))
    lane.set_rightmost_cell(tt(st[0, -1], cuda), tt(st[2, -1], cuda))
    # the float32 glue of set_state_vector_u reproduces the reference's y and u_eq
    assert ulp_diff(lane.get_state_vector()[1].cpu().numpy(), st[1, 1:-1]).max() <= 1
    cr, cy = lane.vectorize_input()
    cr = cr.detach().clone().requires_grad_(True)
    cy = cy.detach().clone().requires_grad_(True)
    nr, ny = dMacroForwardLayer.apply(lane, cr, cy, c["dt"])
    assert rel_max(nr.detach().cpu().numpy(), g[name + "_nr"]) <= 2e-7
    assert rel_max(ny.detach().cpu().numpy(), g[name + "_ny"]) <= 2e-7
    dqs = lane.d_lane[-1].dqs
    assert dqs.shape == (c["N"], 3, 2, 2) and rel_max(dqs, g[name + "_dqs"]) <= 1e-6
    (nr * tt(g[name + "_g_nr"], cuda) + ny * tt(g[name + "_g_ny"], cuda)).sum().backward()
    assert rel_max(cr.grad.cpu().numpy(), g[name + "_g_r"]) <= 1e-6
    assert rel_max(cy.grad.cpu().numpy(), g[name + "_g_y"]) <= 1e-6
    # per-cell view access keeps working
    lane.set_next_state_vector_y(nr, ny)
    lane.update_state()
    assert abs(float(lane.curr_cell[0].state.q.r) - float(g[name + "_nr"][0])) <= 1e-6
    assert abs(float(lane.curr_cell[-1].state.u) - float(g[name + "_nu"][-1])) <= 1e-4


@pytest.mark.parametrize("name", ["small", "sanity"])
def test_road_network_macro_rollout_like_example(cuda, golden_dir, name):
    """example/inverse/macro.py's loop: one dMacroLane in a RoadNetwork, T x network.forward, loss, backward (G4)."""
    import torch
    from road.lane.dmacro_lane import dMacroLane
    from road.network.road_network import RoadNetwork
    g = load(golden_dir, "macro_rollout_%s.npz" % name)
    m = meta_of(g)
    r0, u0 = tt(g["r0"], cuda, True), tt(g["u0"], cuda, True)
    gr, gu = tt(g["ghost_r"], cuda, True), tt(g["ghost_u"], cuda, True)
    lane = dMacroLane(0, m["N"] * m["dx"], m["u_max"], m["dx"])
    lane.set_state_vector_u(r0, u0)
    lane.set_leftmost_cell(gr[0], gu[0])
    lane.set_rightmost_cell(gr[1], gu[1])
    net = RoadNetwork(m["u_max"])
    net.add_lane(lane)
    loss = 0
    for t in range(m["T"]):
        net.forward(m["dt"], True)
        if m["tap"] == "every_sum":
            r, y, u = lane.get_state_vector()
            loss = loss + r.sum() + y.sum() + u.sum()
        if t < len(g["steps_r"]):
            r, y, u = lane.get_state_vector()
            assert rel_max(r.detach().cpu().numpy(), g["steps_r"][t]) <= TOL_STATE
            assert rel_max(u.detach().cpu().numpy(), g["steps_u"][t]) <= TOL_STATE
    rT, yT, uT = lane.get_state_vector()
    if m["tap"] == "final_sq":
        loss = (rT ** 2).sum() + (uT ** 2).sum()
    loss.backward()
    assert rel_max(rT.detach().cpu().numpy(), g["rT"]) <= TOL_STATE
    assert rel_max(uT.detach().cpu().numpy(), g["uT"]) <= TOL_STATE
    assert abs(float(loss.detach()) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    assert rel_max(r0.grad.cpu().numpy(), g["g_r0"]) <= TOL_GRAD
    assert rel_max(u0.grad.cpu().numpy(), g["g_u0"]) <= TOL_GRAD
    assert rel_max(gr.grad.cpu().numpy(), g["g_ghost_r"]) <= TOL_GRAD
    assert rel_max(gu.grad.cpu().numpy(), g["g_ghost_u"]) <= TOL_GRAD
    assert len(lane.d_lane) == m["T"]


@pytest.mark.parametrize("name", ["inv10", "rand24"])
def test_road_network_micro_rollout_like_example(cuda, golden_dir, name):
    """example/inverse/micro.py's loop: vehicles added through the network, T x network.forward, backward (G6)."""
    import torch
    from road.lane.dmicro_lane import dMicroLane
    from road.network.road_network import RoadNetwork
    from road.network.route import MicroRoute
    from road.vehicle.micro_vehicle import MicroVehicle
    g = load(golden_dir, "micro_rollout_%s.npz" % name)
    m = meta_of(g)
    p0, v0 = tt(g["p0"], cuda, True), tt(g["v0"], cuda, True)
    lane = dMicroLane(0, 1e10, m["speed_limit"])
    net = RoadNetwork(m["speed_limit"])
    net.add_lane(lane)
    for i in range(m["V"]):
        mv = MicroVehicle.default_micro_vehicle(m["speed_limit"])
        (mv.accel_max, mv.accel_pref, mv.target_speed, mv.min_space, mv.time_pref, mv.length) = (float(x) for x in g["params"][i])
        mv.position, mv.speed = p0[i], v0[i]
        net.add_vehicle(mv, MicroRoute([0]))
    lane.set_state_vector(p0, v0)
    loss = 0
    for t in range(m["T"]):
        net.forward(m["dt"], True)
        assert lane.head_position_delta == 1000 and lane.head_speed_delta == 0
        if m["tap"] == "every_sum":
            p, v = lane.get_state_vector()
            loss = loss + p.sum() + v.sum()
    pT, vT = lane.get_state_vector()
    if m["tap"] == "final_sq":
        loss = 1e-4 * (pT ** 2).sum() + (vT ** 2).sum()
    loss.backward()
    assert rel_max(pT.detach().cpu().numpy(), g["pT"]) <= 1e-6
    assert rel_max(vT.detach().cpu().numpy(), g["vT"]) <= 1e-6
    assert rel_max(p0.grad.cpu().numpy(), g["g_p0"]) <= 1e-5
    assert rel_max(v0.grad.cpu().numpy(), g["g_v0"]) <= 1e-5
    assert abs(float(lane.curr_vehicle[0].position) - float(g["pT"][0])) <= 1e-3
    assert lane.d_lane[-1].dqs.shape == (m["V"], 2, 2, 2)


def test_scalar_model_surface(cuda, golden_dir):
    """ARZ.riemann_solve / dARZ.compute_dLdR / dARZ.flux_prime / IDM.compute_acceleration on single inputs."""
    from model.macro._arz import ARZ
    from model.macro.darz import dARZ
    from model.micro._idm import IDM
    from model.micro.didm import dIDM
    g = load(golden_dir, "riemann_kat.npz")
    for i in (0, 400, 800, 1200, 1600, 2000):
        row = g["inp"][i]
        um = float(row[8])
        QL, QR = ARZ.FullQ(um), ARZ.FullQ(um)
        QL.q.r, QL.q.y, QL.u, QL.u_eq = (float(x) for x in row[0:4])
        QR.q.r, QR.q.y, QR.u, QR.u_eq = (float(x) for x in row[4:8])
        rs = ARZ.riemann_solve(QL, QR, um)
        assert rs.case_ind == int(g["case"][i])
        assert abs(rs.Q_0.q.r - g["q0"][i, 0]) <= 1e-12 and abs(rs.Q_0.u - g["q0"][i, 2]) <= 1e-10
        assert abs(rs.speed0 - g["speed"][i, 0]) <= 1e-12 * max(1.0, abs(g["speed"][i, 0]))       # both wave speeds, as the
        assert abs(rs.speed1 - g["speed"][i, 1]) <= 1e-12 * max(1.0, abs(g["speed"][i, 1]))       # reference returns them
        dL, dR = dARZ.compute_dLdR(rs, QL, QR, um)
        assert rel_max(dL, g["dL"][i]) <= 2e-7 and np.abs(dR - g["dR"][i]).max() <= 2e-7 * max(1.0, np.abs(g["dR"][i]).max())
    k = load(golden_dir, "idm_kat.npz")
    for i in (0, 300, 650, 790):
        a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt = (float(x) for x in k["inp"][i])
        acc, s, ca, cs = IDM.compute_acceleration(a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt)
        assert abs(acc - k["acc"][i]) <= 1e-10 * max(1.0, abs(k["acc"][i])) and (int(ca), int(cs)) == tuple(k["flags"][i])
        dE = dIDM.compute_dEgo(a_max, a_pref, v, v_t, dp, dv, s0, Tp, s, dt, ca, cs)
        assert rel_max(dE.numpy(), k["dEgo"][i]) <= 1e-6
    # gaps below 1e-5: the acceleration, the optimal spacing and the clip flags come from the CLAMPED gap, the Jacobians take the
    # raw one beside them (dmicro_lane.py:97) -- the arguments are the caller's, they are not re-derived from one gap
    k = load(golden_dir, "idm_kat_smallgap.npz")
    for i in range(0, len(k["inp"]), 7):
        a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt = (float(x) for x in k["inp"][i])
        acc, s, ca, cs = IDM.compute_acceleration(a_max, a_pref, v, v_t, max(dp, 1e-5), dv, s0, Tp, dt)
        assert abs(acc - k["acc"][i]) <= 1e-10 * max(1.0, abs(k["acc"][i])) and (int(ca), int(cs)) == tuple(k["flags"][i])
        dE = dIDM.compute_dEgo(a_max, a_pref, v, v_t, dp, dv, s0, Tp, s, dt, ca, cs)
        dL = dIDM.compute_dLeading(a_max, a_pref, v, v_t, dp, dv, s0, Tp, s, dt, ca, cs)
        assert rel_max(dE.numpy(), k["dEgo"][i]) <= 1e-6 and rel_max(dL.numpy(), k["dLeading"][i]) <= 1e-6
    # closed forms agree between floats and tensors
    import torch
    assert abs(float(ARZ.compute_u_eq(torch.tensor(0.3), 30.0)) - ARZ.compute_u_eq(0.3, 30.0)) <= 1e-5


def test_two_connected_macro_lanes_exchange_ghosts(cuda):
    """RoadNetwork.setup_macro_boundary: a connected neighbour's edge cell is the ghost (road_network.py:299-387);
    two half lanes stepped through the network equal one long lane."""
    import torch
    from road.lane.dmacro_lane import dMacroLane
    from road.network.road_network import RoadNetwork
    rng = np.random.default_rng(3)
    N, dx, dt, um, T = 40, 5.0, 0.01, 30.0, 15
    r0 = rng.uniform(0.1, 0.9, 2 * N).astype(np.float32)
    u0 = rng.uniform(0.0, um, 2 * N).astype(np.float32)
    whole = dMacroLane(0, 2 * N * dx, um, dx)
    whole.set_state_vector_u(tt(r0, cuda), tt(u0, cuda))
    for lane in (whole,):
        lane.set_leftmost_cell(tt(np.float32(0.3), cuda), tt(np.float32(12.0), cuda))
        lane.set_rightmost_cell(tt(np.float32(0.6), cuda), tt(np.float32(5.0), cuda))
    net1 = RoadNetwork(um)
    net1.add_lane(whole)
    a, b = dMacroLane(0, N * dx, um, dx), dMacroLane(1, N * dx, um, dx)
    a.set_state_vector_u(tt(r0[:N], cuda), tt(u0[:N], cuda))
    b.set_state_vector_u(tt(r0[N:], cuda), tt(u0[N:], cuda))
    a.set_leftmost_cell(tt(np.float32(0.3), cuda), tt(np.float32(12.0), cuda))
    b.set_rightmost_cell(tt(np.float32(0.6), cuda), tt(np.float32(5.0), cuda))
    net2 = RoadNetwork(um)
    net2.add_lane(a)
    net2.add_lane(b)
    net2.connect_lane(0, 1)
    for _ in range(T):
        net1.forward(dt, True)
        net2.forward(dt, True)
    rw, _, uw = whole.get_state_vector()
    ra, _, ua = a.get_state_vector()
    rb, _, ub = b.get_state_vector()
    assert torch.equal(torch.cat([ra, rb]), rw) and torch.equal(torch.cat([ua, ub]), uw)


def test_inverse_examples_reduce_the_error(cuda, tmp_path):
    """examples/inverse_macro.py and inverse_micro.py (the harness counterparts of example/inverse/*.py): Adam on the
    HIP path drives the end error down and writes the reference's "{beg} {end}" log lines."""
    import subprocess
    import sys
    from conftest import ROOT
    for script, extra in (("inverse_macro.py", ["--n_cell", "100", "--n_timestep", "200"]),
                          ("inverse_micro.py", ["--n_vehicle", "10", "--n_timestep", "200"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), "--n_episode", "30", "--n_lane", "3",
                              "--seed", "1", "--run_name", "t"] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        kind = script.split("_")[1].split(".")[0]
        lines = open(tmp_path / "result" / "inverse" / "t" / "gd" / "trial_0.txt").read().split("\n")
        lines = [l.split() for l in lines if l]
        assert len(lines) == 30 and all(len(l) == 2 for l in lines)
        assert float(lines[-1][1]) < 0.7 * float(lines[0][1]), (kind, lines[0], lines[-1])
        import shutil
        shutil.rmtree(tmp_path / "result")
    # the hybrid three-lane problem (example/inverse/hybrid.py): through the fused network kernels (one launch each way per
    # episode) and, --lane_by_lane, through the drop-in classes (one operator call per lane-step): the same log lines
    logs = {}
    for name, extra in (("h", []), ("hl", ["--lane_by_lane"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "inverse_hybrid.py"), "--n_episode", "6", "--n_timestep", "120",
                              "--seed", "3", "--run_name", name] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        print(out.stdout.strip())
        lines = [l.split() for l in open(tmp_path / "result" / "inverse" / name / "gd" / "trial_0.txt").read().split("\n") if l]
        assert len(lines) == 6 and all(len(l) == 2 for l in lines)
        assert float(lines[-1][1]) < float(lines[0][1]), lines
        logs[name] = np.array(lines, dtype=np.float64)
    assert np.allclose(logs["h"], logs["hl"], rtol=2e-4, atol=1e-6), (logs["h"], logs["hl"])


@pytest.mark.parametrize("name", ["hybrid3", "hybrid3_b", "hybrid3_c", "hybrid3_d", "x1_4", "x2_11", "x5_9"])
def test_hybrid_three_lane_network_like_example(cuda, golden_dir, name):
    """example/inverse/hybrid.py's network macro(0) -> micro(1) -> macro(2) through the mirror (G7): flux-capacitor
    spawning, micro -> macro hand-off with the ancillary variable `a`, same event times, states and gradients."""
    import torch
    from road.lane.dmacro_lane import dMacroLane
    from road.lane.dmicro_lane import dMicroLane
    from road.network.road_network import RoadNetwork
    g = load(golden_dir, "hybrid_%s.npz" % name)
    m = meta_of(g)
    N, T, dx, dt, um = m["N"], m["T"], m["dx"], m["dt"], m["u_max"]
    np.random.seed(m["seed"])
    r0, u0 = tt(g["r0"], cuda, True), tt(g["u0"], cuda, True)
    bd_r, bd_u = tt(g["bd_r"], cuda), tt(g["bd_u"], cuda)
    net = RoadNetwork(um)
    a = dMacroLane(0, N * dx, um, dx)
    a.set_leftmost_cell(bd_r[0], bd_u[0])
    a.set_rightmost_cell(bd_r[1], bd_u[1])
    net.add_lane(a)
    a.set_state_vector_u(r0, u0)
    b = dMicroLane(1, N * dx, um)
    net.add_lane(b)
    c = dMacroLane(2, N * dx, um, dx)
    c.set_leftmost_cell(bd_r[2], bd_u[2])
    c.set_rightmost_cell(bd_r[3], bd_u[3])
    net.add_lane(c)
    net.connect_lane(0, 1)
    net.connect_lane(1, 2)
    net.macro_route = net.create_random_macro_route()
    assert sorted(net.macro_route.next_lane_dict.items()) == [tuple(x) for x in g["macro_next"].tolist()]
    events, nveh = [], []
    for t in range(T):
        before, spawned = b.num_vehicle(), net.num_vehicle
        net.forward(dt, True)
        if net.num_vehicle > spawned:
            events.append((t, 0, float(b.curr_vehicle[0].speed)))
        if b.num_vehicle() < before + (net.num_vehicle - spawned):
            events.append((t, 1, float(c.curr_cell[0].state.q.r)))
        nveh.append(b.num_vehicle())
    ref_ev = g["events"]
    assert [(e[0], e[1]) for e in events] == [(int(e[0]), int(e[1])) for e in ref_ev]
    assert np.allclose([e[2] for e in events], ref_ev[:, 2], rtol=1e-5, atol=1e-6)
    assert nveh == g["nveh"].tolist()
    rA, yA, uA = a.get_state_vector()
    rC, yC, uC = c.get_state_vector()
    pB, vB = b.get_state_vector()
    loss = (rC ** 2).sum() + (uC ** 2).sum() + (rA ** 2).sum() + (uA ** 2).sum() + 1e-4 * (pB ** 2).sum() + (vB ** 2).sum()
    loss.backward()
    for got, key in ((rA, "rA"), (uA, "uA"), (rC, "rC"), (uC, "uC"), (pB, "pB"), (vB, "vB")):
        assert rel_max(got.detach().cpu().numpy(), g[key]) <= TOL_STATE, key
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert rel_max(r0.grad.cpu().numpy(), g["g_r0"]) <= TOL_GRAD
    assert rel_max(u0.grad.cpu().numpy(), g["g_u0"]) <= TOL_GRAD


def test_macro_state_of_micro_lane_like_reference(cuda, golden_dir):
    """RoadNetwork.get_macro_state_of_micro_lane (reference road_network.py:207-297) on micro -> micro -> micro chains:
    vehicles on the lane, on the upstream lane routed onto it and on the downstream lane that came through it, soft
    (sigmoid) and hard weights, against the reference's own values."""
    import json
    from road.lane._micro_lane import MicroLane
    from road.network.road_network import RoadNetwork
    from road.network.route import MicroRoute
    from road.vehicle.micro_vehicle import MicroVehicle
    g = load(golden_dir, "macro_state_of_micro_lane.npz")
    for case in json.loads(str(g["cases"])):
        sl = 30.0
        net = RoadNetwork(sl)
        for i, ln in enumerate(case["lengths"]):
            net.add_lane(MicroLane(i, ln, sl))
        net.connect_lane(0, 1)
        net.connect_lane(1, 2)
        for lane_id, pos, spd, route, idx in case["vehicles"]:
            mv = MicroVehicle.default_micro_vehicle(sl)
            mv.position, mv.speed = pos, spd
            r = MicroRoute(list(route))
            for _ in range(idx):
                r.increment_curr_idx()
            net.add_vehicle(mv, r)
        for flag, key in ((True, "soft"), (False, "hard")):
            d, s = net.get_macro_state_of_micro_lane(1, flag)
            assert abs(float(d) - case[key][0]) <= 1e-6 * max(1.0, abs(case[key][0])), (key, float(d), case[key])
            assert abs(float(s) - case[key][1]) <= 1e-5 * max(1.0, abs(case[key][1])), (key, float(s), case[key])
