#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference in this container.

Test tooling only.  Run here (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens.py [--only G1,G4 ...]

It imports the reference's own implementation of the hot path from /root/reference
(model.macro._arz / darz, model.micro._idm / didm, road.lane.dmacro_lane / dmicro_lane,
road.network.road_network) and records, as plain arrays in tests/golden/*.npz, the inputs
it fed and the outputs the reference produced.  Nothing from the reference's source text
is stored: the fixtures hold data only.  Seeds and library versions are stored in each
file under the key `meta`.

Golden sets:
  G1/G2  riemann_kat.npz     ARZ.riemann_solve + dARZ.compute_dLdR + dARZ.flux_prime per interface
  G3     macro_step.npz      one dMacroLane step: next state, Jacobian tape dqs, backward of cotangents
  G4     macro_rollout_*.npz T-step rollouts through RoadNetwork.forward with loss and gradients
  G5     idm_kat.npz         IDM.compute_acceleration + dIDM.compute_dEgo/dLeading
  G6     micro_rollout_*.npz dMicroLane rollouts with loss and gradients
  G7     hybrid_hybrid3.npz  macro -> micro -> macro network with spawn / hand-off events, loss and gradients
  G8     itscp_*.npz         itscp environment (needs tools/ref_stubs for highway_env / gym / pygame): lane table,
                             schedules, per-step macro routes, action -> reward, d reward / d action, per-step queues
"""
import argparse
import json
import os
import sys
import time

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REFERENCE = "/root/reference"
sys.path.insert(0, REFERENCE)

import numpy as np  # noqa: E402
import torch as th  # noqa: E402

from model.macro._arz import ARZ, EPSILON  # noqa: E402
from model.macro.darz import dARZ  # noqa: E402
from model.micro._idm import IDM  # noqa: E402
from model.micro.didm import dIDM  # noqa: E402
from road.lane.dmacro_lane import dMacroLane, dMacroForwardLayer  # noqa: E402
from road.lane.dmicro_lane import dMicroLane  # noqa: E402
from road.network.road_network import RoadNetwork  # noqa: E402
from road.network.route import MicroRoute  # noqa: E402
from road.vehicle.micro_vehicle import MicroVehicle  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def meta(**kw):
    d = dict(torch=th.__version__, numpy=np.__version__, python=sys.version.split()[0],
             generator="tools/gen_goldens.py", reference="SonSang/diff-hybrid-traffic-sim @ /root/reference")
    d.update(kw)
    return np.array(json.dumps(d))


# ----------------------------------------------------------------------------------------------
# G1/G2: interface KATs
# ----------------------------------------------------------------------------------------------

def fullq_from_r_u_f32(r, u, u_max):
    """Build a cell state the way lanes do for initial/ghost cells: float32 torch glue
    (FullQ.from_r_u on 0-dim tensors) then detach to Python floats (dMacroLane.decell)."""
    fq = ARZ.FullQ.from_r_u(th.tensor(r, dtype=th.float32), th.tensor(u, dtype=th.float32), u_max)
    return detach(fq, u_max)


def fullq_from_r_y_f32(r, y, u_max):
    """Interior cells after a step: FullQ.set_r_y on 0-dim float32 tensors, then detach."""
    fq = ARZ.FullQ(u_max)
    fq.set_r_y(th.tensor(r, dtype=th.float32), th.tensor(y, dtype=th.float32), u_max)
    return detach(fq, u_max)


def detach(fq, u_max):
    out = ARZ.FullQ(u_max)
    out.q.r = float(fq.q.r)
    out.q.y = float(fq.q.y)
    out.u = float(fq.u)
    out.u_eq = float(fq.u_eq)
    return out


def branch_of(QL, QR, u_max):
    """Which of the six branches of the solver the pair takes (numbering of SURVEY 8a A3)."""
    if QL.q.r < EPSILON:
        return 1
    if QR.q.r < EPSILON:
        return 2
    if abs(QL.u - QR.u) < EPSILON:
        return 3
    if QL.u > QR.u:
        return 4
    if u_max + QL.u - QL.u_eq > QR.u:
        return 5
    return 6


def gen_riemann_kat(seed=20261002):
    rng = np.random.default_rng(seed)
    pairs = []  # (QL, QR, u_max)

    def add(QL, QR, um):
        pairs.append((QL, QR, um))

    for um in (30.0, 13.5, 20.0):
        # generic random pairs built from (r, u)
        for _ in range(260):
            rl, rr = rng.uniform(0.0, 1.0, 2)
            ul, ur = rng.uniform(0.0, um, 2)
            add(fullq_from_r_u_f32(rl, ul, um), fullq_from_r_u_f32(rr, ur, um), um)
        # interior-like cells built from (r, y), including r > 1 and negative y
        for _ in range(120):
            rl, rr = rng.uniform(0.0, 1.3, 2)
            yl, yr = rng.uniform(-6.0, 6.0, 2)
            add(fullq_from_r_y_f32(rl, yl, um), fullq_from_r_y_f32(rr, yr, um), um)
        # near-equilibrium pairs (u = u_eq(r)): small y
        for _ in range(40):
            rl, rr = rng.uniform(0.02, 0.98, 2)
            ul = float(ARZ.compute_u_eq(rl, um))
            ur = float(ARZ.compute_u_eq(rr, um))
            add(fullq_from_r_u_f32(rl, ul, um), fullq_from_r_u_f32(rr, ur, um), um)
        # vacuum on the left / right / both
        for rv in (0.0, 1e-6, 5e-6, 9.9e-6):
            for _ in range(6):
                r2 = rng.uniform(0.05, 1.0)
                u1, u2 = rng.uniform(0.0, um, 2)
                add(fullq_from_r_u_f32(rv, u1, um), fullq_from_r_u_f32(r2, u2, um), um)
                add(fullq_from_r_u_f32(r2, u2, um), fullq_from_r_u_f32(rv, u1, um), um)
            add(fullq_from_r_u_f32(rv, um, um), fullq_from_r_u_f32(rv, um, um), um)
        # right vacuum with negative lambda_0 on the left (dense, slow left state) -> case 2
        for _ in range(20):
            rl = rng.uniform(0.5, 1.0)
            ul = rng.uniform(0.0, 0.3 * um)
            add(fullq_from_r_u_f32(rl, ul, um), fullq_from_r_u_f32(0.0, um, um), um)
        # equal speeds (branch 3): same u on both sides, and |du| just under/over epsilon
        for _ in range(20):
            rl, rr = rng.uniform(0.05, 1.0, 2)
            u = float(np.float32(rng.uniform(0.0, um)))
            add(fullq_from_r_u_f32(rl, u, um), fullq_from_r_u_f32(rr, u, um), um)
        # slow dense right state: shocks with negative speed (branch 4 case 1)
        for _ in range(60):
            rl = rng.uniform(0.3, 1.0)
            rr = rng.uniform(0.6, 1.0)
            ul = rng.uniform(0.2 * um, um)
            ur = rng.uniform(0.0, 0.2 * um)
            add(fullq_from_r_u_f32(rl, ul, um), fullq_from_r_u_f32(rr, ur, um), um)
        # rarefactions: dense slow left, faster right (branch 5 all cases, branch 6)
        for _ in range(120):
            rl = rng.uniform(0.2, 1.0)
            ul = rng.uniform(0.0, 0.5 * um)
            rr = rng.uniform(0.01, 1.0)
            ur = ul + rng.uniform(0.0, 1.2 * um)
            add(fullq_from_r_u_f32(rl, ul, um), fullq_from_r_u_f32(rr, ur, um), um)

    n = len(pairs)
    inp = np.zeros((n, 9), dtype=np.float64)  # rL yL uL ueqL rR yR uR ueqR u_max
    case = np.zeros(n, dtype=np.int32)
    branch = np.zeros(n, dtype=np.int32)
    q0 = np.zeros((n, 4), dtype=np.float64)   # r y u u_eq of Q_0
    speed = np.zeros((n, 2), dtype=np.float64)
    dL = np.zeros((n, 2, 2), dtype=np.float32)
    dR = np.zeros((n, 2, 2), dtype=np.float32)
    fp = np.zeros((n, 2, 2), dtype=np.float32)
    for i, (QL, QR, um) in enumerate(pairs):
        inp[i] = (QL.q.r, QL.q.y, QL.u, QL.u_eq, QR.q.r, QR.q.y, QR.u, QR.u_eq, um)
        rs = ARZ.riemann_solve(QL, QR, um)
        case[i] = rs.case_ind
        branch[i] = branch_of(QL, QR, um)
        q0[i] = (rs.Q_0.q.r, rs.Q_0.q.y, rs.Q_0.u, rs.Q_0.u_eq)
        speed[i] = (rs.speed0, rs.speed1)
        a, b = dARZ.compute_dLdR(rs, QL, QR, um)
        dL[i], dR[i] = a, b
        fp[i] = dARZ.flux_prime(rs.Q_0)
    combos = sorted(set(zip(branch.tolist(), case.tolist())))
    print("G1/G2: %d pairs, (branch,case) combos hit: %s" % (n, combos))
    assert np.all(np.isfinite(q0)) and np.all(np.isfinite(dL)) and np.all(np.isfinite(dR))
    np.savez_compressed(os.path.join(OUT, "riemann_kat.npz"), inp=inp, case=case, branch=branch, q0=q0,
                        speed=speed, dL=dL, dR=dR, fp=fp, meta=meta(seed=seed, combos=combos))


def gen_riemann_kat_stale(seed=20261003):
    """Interface KATs on states as Conversion.micro_to_macro leaves them (conversion.py:124-166): r, u, y rewritten
    from the deposited vehicle while the cell's u_eq still belongs to the density before the deposit."""
    rng = np.random.default_rng(seed)
    pairs = []

    def deposited(um):
        r_old = float(rng.choice([0.0, 1e-6, rng.uniform(0.0, 0.4), rng.uniform(0.0, 1.0)]))
        fq = fullq_from_r_u_f32(r_old, rng.uniform(0.0, um), um)
        n_r = th.tensor(min(max(r_old + rng.uniform(0.0, 1.0), 1e-5), 1.0 - 1e-5), dtype=th.float32)
        speed = th.tensor(rng.uniform(0.0, um), dtype=th.float32)
        fq.q.r = float(n_r)
        fq.u = float(speed)
        fq.q.y = float(ARZ.compute_y(n_r, speed, um))
        return fq

    def plain(um):
        if rng.uniform() < 0.5:
            return fullq_from_r_u_f32(rng.uniform(0.0, 1.0), rng.uniform(0.0, um), um)
        return fullq_from_r_y_f32(rng.uniform(0.0, 1.2), rng.uniform(-4.0, 4.0), um)

    for um in (60.0, 30.0, 13.5):
        for _ in range(200):
            pairs.append((deposited(um), plain(um), um))
            pairs.append((plain(um), deposited(um), um))
            pairs.append((deposited(um), deposited(um), um))
    n = len(pairs)
    inp = np.zeros((n, 9), dtype=np.float64)
    case = np.zeros(n, dtype=np.int32)
    branch = np.zeros(n, dtype=np.int32)
    q0 = np.zeros((n, 4), dtype=np.float64)
    speed = np.zeros((n, 2), dtype=np.float64)
    dL = np.zeros((n, 2, 2), dtype=np.float32)
    dR = np.zeros((n, 2, 2), dtype=np.float32)
    fp = np.zeros((n, 2, 2), dtype=np.float32)
    for i, (QL, QR, um) in enumerate(pairs):
        inp[i] = (QL.q.r, QL.q.y, QL.u, QL.u_eq, QR.q.r, QR.q.y, QR.u, QR.u_eq, um)
        rs = ARZ.riemann_solve(QL, QR, um)
        case[i] = rs.case_ind
        branch[i] = branch_of(QL, QR, um)
        q0[i] = (rs.Q_0.q.r, rs.Q_0.q.y, rs.Q_0.u, rs.Q_0.u_eq)
        speed[i] = (rs.speed0, rs.speed1)
        a, b = dARZ.compute_dLdR(rs, QL, QR, um)
        dL[i], dR[i] = a, b
        fp[i] = dARZ.flux_prime(rs.Q_0)
    combos = sorted(set(zip(branch.tolist(), case.tolist())))
    print("G1s: %d stale-u_eq pairs, combos %s" % (n, combos))
    np.savez_compressed(os.path.join(OUT, "riemann_kat_stale.npz"), inp=inp, case=case, branch=branch, q0=q0,
                        speed=speed, dL=dL, dR=dR, fp=fp, meta=meta(seed=seed, combos=combos))


# ----------------------------------------------------------------------------------------------
# G3: one dMacroLane step
# ----------------------------------------------------------------------------------------------

def lane_padded_state(lane):
    """(r, y, u, u_eq)[N+2] as float32 values: ghosts at both ends."""
    cells = [lane.leftmost_cell] + list(lane.curr_cell) + [lane.rightmost_cell]
    out = np.zeros((4, len(cells)), dtype=np.float32)
    for i, c in enumerate(cells):
        out[0, i] = float(c.state.q.r)
        out[1, i] = float(c.state.q.y)
        out[2, i] = float(c.state.u)
        out[3, i] = float(c.state.u_eq)
    return out


def gen_macro_step(seed=7):
    th.manual_seed(seed)
    cases = {}
    configs = [
        # name, N, dx, dt, u_max, kind
        ("rand64", 64, 5.0, 0.01, 30.0, "rand"),
        ("sanity100", 100, 100.0, 0.03, 30.0, "sanity"),
        ("vacuum9", 9, 5.0, 0.01, 30.0, "vacuum"),
        ("single1", 1, 5.0, 0.01, 30.0, "rand"),
        ("jam33", 33, 4.0, 0.02, 20.0, "jam"),
    ]
    for name, N, dx, dt, um, kind in configs:
        if kind == "rand":
            r = th.rand(N + 2)
            u = th.rand(N + 2) * um
        elif kind == "sanity":
            r = th.rand(N + 2)
            u = th.lerp(th.tensor([0.4 * um]), th.tensor([0.7 * um]), th.rand(N + 2))
        elif kind == "vacuum":
            r = th.tensor([0.3, 0.0, 0.0, 0.5, 1e-6, 0.7, 0.0, 0.2, 0.9, 0.0, 0.4])
            u = th.tensor([10.0, 30.0, 30.0, 5.0, 30.0, 2.0, 30.0, 25.0, 1.0, 30.0, 12.0])
        elif kind == "jam":
            r = th.cat([th.rand(17) * 0.3, 0.7 + th.rand(N + 2 - 17) * 0.3])
            u = th.cat([15 + th.rand(17) * 5.0, th.rand(N + 2 - 17) * 2.0])
        r = r.to(th.float32)
        u = u.to(th.float32)
        lane = dMacroLane(0, N * dx, um, dx)
        lane.set_state_vector_u(r[1:-1], u[1:-1])
        lane.set_leftmost_cell(r[0], u[0])
        lane.set_rightmost_cell(r[-1], u[-1])
        state = lane_padded_state(lane)
        cr, cy = lane.vectorize_input()
        cr = cr.detach().clone().requires_grad_(True)
        cy = cy.detach().clone().requires_grad_(True)
        nr, ny = dMacroForwardLayer.apply(lane, cr, cy, dt)
        dqs = lane.d_lane[-1].dqs.copy()
        rs = lane.riemann_solution
        case = np.array([s.case_ind for s in rs], dtype=np.int32)
        speeds = np.array([[s.speed0, s.speed1] for s in rs], dtype=np.float64)
        # the float32 glue for the next step's u, u_eq (set_next_state_vector_y)
        lane.set_next_state_vector_y(nr.detach(), ny.detach())
        nu = np.array([float(c.state.u) for c in lane.next_cell], dtype=np.float32)
        nueq = np.array([float(c.state.u_eq) for c in lane.next_cell], dtype=np.float32)
        g_nr = th.randn(N)
        g_ny = th.randn(N)
        (nr * g_nr + ny * g_ny).sum().backward()
        cases[name] = dict(N=N, dx=dx, dt=dt, u_max=um)
        pre = name + "_"
        cases[pre + "state"] = state
        cases[pre + "nr"] = nr.detach().numpy().copy()
        cases[pre + "ny"] = ny.detach().numpy().copy()
        cases[pre + "nu"] = nu
        cases[pre + "nueq"] = nueq
        cases[pre + "dqs"] = dqs
        cases[pre + "case"] = case
        cases[pre + "speed"] = speeds
        cases[pre + "g_nr"] = g_nr.numpy().copy()
        cases[pre + "g_ny"] = g_ny.numpy().copy()
        cases[pre + "g_r"] = cr.grad.numpy().copy()
        cases[pre + "g_y"] = cy.grad.numpy().copy()
        print("G3 %-10s N=%3d cases=%s" % (name, N, np.bincount(case, minlength=3).tolist()))
    cfg = {k: v for k, v in cases.items() if isinstance(v, dict)}
    arrays = {k: v for k, v in cases.items() if not isinstance(v, dict)}
    np.savez_compressed(os.path.join(OUT, "macro_step.npz"), meta=meta(seed=seed, configs=cfg), **arrays)


# ----------------------------------------------------------------------------------------------
# G4: macro rollouts through RoadNetwork.forward
# ----------------------------------------------------------------------------------------------

def macro_rollout(name, N, T, dx, dt, um, seed, init="uniform", tap="final_sq", record_steps=0):
    """One straight dMacroLane in a RoadNetwork (example/inverse/macro.py:34-68 construction),
    T x RoadNetwork.forward(dt, True), loss on (r_T, u_T), backward to r0, u0 and ghost (r, u)."""
    th.manual_seed(seed)
    if init == "uniform":       # section 8d C1: r0 = rand, u0 = rand * u_max, ghosts likewise
        r0 = th.rand(N)
        u0 = th.rand(N) * um
        gr = th.rand(2)
        gu = th.rand(2) * um
    elif init == "bench":       # section 8d C2 regime: r0 in [0.05, 0.95]
        r0 = 0.05 + 0.9 * th.rand(N)
        u0 = th.rand(N) * um
        gr = 0.05 + 0.9 * th.rand(2)
        gu = th.rand(2) * um
    elif init == "sanity":      # example/sanity/macro.py regime
        r0 = th.rand(N)
        u0 = th.lerp(th.tensor([0.4 * um]), th.tensor([0.7 * um]), th.rand(N))
        gr = th.rand(2)
        gu = th.lerp(th.tensor([0.4 * um]), th.tensor([0.7 * um]), th.rand(2))
    r0 = r0.to(th.float32).requires_grad_(True)
    u0 = u0.to(th.float32).requires_grad_(True)
    gr = gr.to(th.float32).requires_grad_(True)
    gu = gu.to(th.float32).requires_grad_(True)

    lane = dMacroLane(0, N * dx, um, dx)
    lane.set_state_vector_u(r0, u0)
    lane.set_leftmost_cell(gr[0], gu[0])
    lane.set_rightmost_cell(gr[1], gu[1])
    net = RoadNetwork(um)
    net.add_lane(lane)

    steps_r, steps_y, steps_u = [], [], []
    loss = 0
    t0 = time.time()
    for t in range(T):
        net.forward(dt, True)
        if tap == "every_sum":      # sanity script: sum of r, y, u after every step
            r, y, u = lane.get_state_vector()
            loss = loss + r.sum() + y.sum() + u.sum()
        if t < record_steps:
            r, y, u = lane.get_state_vector()
            steps_r.append(r.detach().numpy().copy())
            steps_y.append(y.detach().numpy().copy())
            steps_u.append(u.detach().numpy().copy())
    rT, yT, uT = lane.get_state_vector()
    if tap == "final_sq":
        loss = (rT ** 2).sum() + (uT ** 2).sum()
    t1 = time.time()
    loss.backward()
    t2 = time.time()
    case_hist = np.zeros(3, dtype=np.int64)
    print("G4 %-12s N=%d T=%d loss=%.6f fwd %.1fs bwd %.1fs" % (name, N, T, float(loss), t1 - t0, t2 - t1))
    np.savez_compressed(
        os.path.join(OUT, "macro_rollout_%s.npz" % name),
        r0=r0.detach().numpy(), u0=u0.detach().numpy(), ghost_r=gr.detach().numpy(), ghost_u=gu.detach().numpy(),
        rT=rT.detach().numpy(), yT=yT.detach().numpy(), uT=uT.detach().numpy(),
        loss=np.float64(float(loss)),
        g_r0=r0.grad.numpy(), g_u0=u0.grad.numpy(), g_ghost_r=gr.grad.numpy(), g_ghost_u=gu.grad.numpy(),
        steps_r=np.array(steps_r, dtype=np.float32), steps_y=np.array(steps_y, dtype=np.float32),
        steps_u=np.array(steps_u, dtype=np.float32),
        meta=meta(seed=seed, N=N, T=T, dx=dx, dt=dt, u_max=um, init=init, tap=tap,
                  ref_seconds_fwd=t1 - t0, ref_seconds_bwd=t2 - t1))


def gen_macro_rollouts(which):
    if "c1" in which:        # BASELINE config #1: 1 lane x 100 cells x 200 steps
        macro_rollout("c1", 100, 200, 5.0, 0.01, 30.0, seed=1, init="uniform", tap="final_sq", record_steps=8)
    if "sanity" in which:    # example/sanity/macro.py regime: 100 cells x 10 steps, dx=100, dt=0.03
        macro_rollout("sanity", 100, 10, 100.0, 0.03, 30.0, seed=0, init="sanity", tap="every_sum", record_steps=10)
    if "small" in which:     # quick: 24 cells x 40 steps
        macro_rollout("small", 24, 40, 5.0, 0.01, 30.0, seed=3, init="uniform", tap="final_sq", record_steps=40)
    if "bench64" in which:   # C2 regime at oracle-checkable size: 64 cells x 300 steps
        macro_rollout("bench64", 64, 300, 5.0, 0.01, 30.0, seed=2026, init="bench", tap="final_sq", record_steps=4)
    if "long" in which:      # error growth check at T = 1000
        macro_rollout("long", 48, 1000, 5.0, 0.01, 30.0, seed=11, init="bench", tap="final_sq", record_steps=0)
    if "c2slice" in which:   # ONE lane of BASELINE config 2 at its full shape: 512 cells x 1000 steps (round 3: pins the
        # very kernel instantiations bench.py times -- four wavefronts x two passes, no history -- against the reference)
        macro_rollout("c2slice", 512, 1000, 5.0, 0.01, 30.0, seed=2027, init="bench", tap="final_sq", record_steps=0)


# ----------------------------------------------------------------------------------------------
# G5: IDM KATs
# ----------------------------------------------------------------------------------------------

def gen_idm_kat(seed=5):
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(600):
        sl = rng.choice([30.0, 20.0, 13.5])
        a_max = sl * rng.uniform(0.5, 1.5)
        a_pref = sl * rng.uniform(0.5, 1.5)
        v_t = sl * rng.uniform(0.8, 1.2)
        s0 = rng.uniform(0.5, 5.0)
        T = rng.uniform(0.1, 3.0)
        v = float(np.float32(rng.uniform(0.0, 1.2 * sl)))
        dp = float(np.float32(rng.choice([rng.uniform(0.5, 50.0), rng.uniform(1e-5, 0.5), 1000.0])))
        dv = float(np.float32(rng.uniform(-sl, sl)))
        dt = rng.choice([0.01, 1.0 / 30.0, 0.1])
        rows.append((a_max, a_pref, v, v_t, dp, dv, s0, T, dt))
    # force negative optimal spacing (leader much faster) and acceleration clipping (tiny gap)
    for _ in range(60):
        rows.append((30.0, 24.0, float(np.float32(rng.uniform(5, 30))), 27.0, float(np.float32(rng.uniform(5, 100))),
                     float(np.float32(-rng.uniform(20, 60))), 0.5, 0.1, 0.01))
        rows.append((30.0, 24.0, float(np.float32(rng.uniform(0.0, 3.0))), 27.0, float(np.float32(rng.uniform(1e-5, 0.2))),
                     float(np.float32(rng.uniform(0, 5))), 0.5, 0.1, 0.01))
    # default vehicle (road/vehicle/micro_vehicle.py:31-72) at speed limit 30
    for _ in range(80):
        rows.append((30.0 * 1.0, 30.0 * 0.8, float(np.float32(rng.uniform(0, 30))), 30.0 * 0.9,
                     float(np.float32(rng.uniform(0.1, 60))), float(np.float32(rng.uniform(-10, 10))), 5.0 * 0.1, 0.1, 0.01))
    inp = np.array(rows, dtype=np.float64)
    n = len(rows)
    acc = np.zeros(n)
    sstar = np.zeros(n)
    flags = np.zeros((n, 2), dtype=np.int32)
    dE = np.zeros((n, 2, 2), dtype=np.float32)
    dLd = np.zeros((n, 2, 2), dtype=np.float32)
    for i, (a_max, a_pref, v, v_t, dp, dv, s0, T, dt) in enumerate(rows):
        a, s, ca, cs = IDM.compute_acceleration(a_max, a_pref, v, v_t, dp, dv, s0, T, dt)
        acc[i], sstar[i] = a, s
        flags[i] = (int(ca), int(cs))
        dE[i] = dIDM.compute_dEgo(a_max, a_pref, v, v_t, dp, dv, s0, T, s, dt, ca, cs).numpy()
        dLd[i] = dIDM.compute_dLeading(a_max, a_pref, v, v_t, dp, dv, s0, T, s, dt, ca, cs).numpy()
    print("G5: %d rows; clipped_acc %d, clipped_spacing %d" % (n, flags[:, 0].sum(), flags[:, 1].sum()))
    np.savez_compressed(os.path.join(OUT, "idm_kat.npz"), inp=inp, acc=acc, sstar=sstar, flags=flags,
                        dEgo=dE, dLeading=dLd, meta=meta(seed=seed, columns="a_max a_pref v v_target dp dv min_space time_pref dt"))


def gen_idm_kat_smallgap(seed=11):
    """G5b: vehicles whose raw gap is below POSITION_DELTA_EPS = 1e-5.  MicroLane.forward clamps the gap for the acceleration
    (_micro_lane.py:166) and dMicroLane._backward hands the UN-clamped gap to the Jacobians together with the optimal spacing and
    the clip flags of that forward call (dmicro_lane.py:97): the arguments of dIDM.compute_dEgo / compute_dLeading do not all
    come from one gap."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(60):
        sl = rng.choice([30.0, 20.0])
        rows.append((sl * 1.0, sl * 0.8, float(np.float32(rng.uniform(0.0, 0.5 * sl))), sl * 0.9, float(np.float32(rng.uniform(1e-7, 9.9e-6))),
                     float(np.float32(rng.uniform(-5, 5))), 0.5, 0.1, rng.choice([0.01, 1.0 / 30.0])))
    for _ in range(20):     # leader much faster: negative optimal spacing, the acceleration is not clipped
        rows.append((30.0, 24.0, float(np.float32(rng.uniform(1, 20))), 27.0, float(np.float32(rng.uniform(1e-7, 9.9e-6))),
                     float(np.float32(-rng.uniform(20, 60))), 0.5, 0.1, 0.01))
    inp = np.array(rows, dtype=np.float64)
    n = len(rows)
    acc, sstar = np.zeros(n), np.zeros(n)
    flags = np.zeros((n, 2), dtype=np.int32)
    dE, dLd = np.zeros((n, 2, 2), dtype=np.float32), np.zeros((n, 2, 2), dtype=np.float32)
    for i, (a_max, a_pref, v, v_t, dp, dv, s0, T, dt) in enumerate(rows):
        a, s, ca, cs = IDM.compute_acceleration(a_max, a_pref, v, v_t, max(dp, 1e-5), dv, s0, T, dt)
        acc[i], sstar[i] = a, s
        flags[i] = (int(ca), int(cs))
        dE[i] = dIDM.compute_dEgo(a_max, a_pref, v, v_t, dp, dv, s0, T, s, dt, ca, cs).numpy()
        dLd[i] = dIDM.compute_dLeading(a_max, a_pref, v, v_t, dp, dv, s0, T, s, dt, ca, cs).numpy()
    print("G5b: %d rows; clipped_acc %d, clipped_spacing %d" % (n, flags[:, 0].sum(), flags[:, 1].sum()))
    np.savez_compressed(os.path.join(OUT, "idm_kat_smallgap.npz"), inp=inp, acc=acc, sstar=sstar, flags=flags, dEgo=dE, dLeading=dLd,
                        meta=meta(seed=seed, columns="a_max a_pref v v_target dp_raw dv min_space time_pref dt",
                                  note="acc / sstar / flags from max(dp_raw, 1e-5); Jacobians from dp_raw with that sstar and those flags"))


# ----------------------------------------------------------------------------------------------
# G6: dMicroLane rollouts through RoadNetwork.forward
# ----------------------------------------------------------------------------------------------

def micro_rollout(name, V, T, dt, sl, seed, params="default", tap="final_sq", head=(1000, 0), record_steps=0,
                  spacing=4.0, jitter=2.0, vlo=0.3, vhi=0.7):
    th.manual_seed(seed)
    rng = np.random.default_rng(seed)
    vlen = 5.0
    p0 = th.arange(0, V) * spacing * vlen + th.rand(V) * jitter * vlen       # example/inverse/micro.py:78-80
    v0 = th.lerp(th.tensor([vlo * sl]), th.tensor([vhi * sl]), th.rand(V))
    p0 = p0.to(th.float32).requires_grad_(True)
    v0 = v0.to(th.float32).requires_grad_(True)
    lane = dMicroLane(0, 1e10, sl)
    net = RoadNetwork(sl)
    net.add_lane(lane)
    par = np.zeros((V, 6), dtype=np.float64)  # a_max a_pref v_target min_space time_pref length
    for i in range(V):
        mv = MicroVehicle.default_micro_vehicle(sl)
        if params == "random":
            mv.accel_max = float(sl * rng.uniform(0.8, 1.5))
            mv.accel_pref = float(sl * rng.uniform(0.6, 1.5))
            mv.target_speed = float(sl * rng.uniform(0.8, 1.2))
            mv.min_space = float(vlen * rng.uniform(0.1, 1.0))
            mv.time_pref = float(rng.uniform(0.1, 1.5))
        mv.position = p0[i]
        mv.speed = v0[i]
        par[i] = (mv.accel_max, mv.accel_pref, mv.target_speed, mv.min_space, mv.time_pref, mv.length)
        net.add_vehicle(mv, MicroRoute([0]))
    lane.set_state_vector(p0, v0)
    steps_p, steps_v = [], []
    loss = 0
    t0 = time.time()
    for t in range(T):
        net.forward(dt, True)
        assert lane.head_position_delta == head[0] and lane.head_speed_delta == head[1]
        if tap == "every_sum":
            p, v = lane.get_state_vector()
            loss = loss + p.sum() + v.sum()
        if t < record_steps:
            p, v = lane.get_state_vector()
            steps_p.append(p.detach().numpy().copy())
            steps_v.append(v.detach().numpy().copy())
    pT, vT = lane.get_state_vector()
    if tap == "final_sq":
        loss = 1e-4 * (pT ** 2).sum() + (vT ** 2).sum()      # section 8d C3 loss taps
    t1 = time.time()
    loss.backward()
    t2 = time.time()
    print("G6 %-10s V=%d T=%d loss=%.6f fwd %.1fs bwd %.1fs" % (name, V, T, float(loss), t1 - t0, t2 - t1))
    np.savez_compressed(
        os.path.join(OUT, "micro_rollout_%s.npz" % name),
        p0=p0.detach().numpy(), v0=v0.detach().numpy(), params=par, pT=pT.detach().numpy(), vT=vT.detach().numpy(),
        loss=np.float64(float(loss)), g_p0=p0.grad.numpy(), g_v0=v0.grad.numpy(),
        steps_p=np.array(steps_p, dtype=np.float32), steps_v=np.array(steps_v, dtype=np.float32),
        meta=meta(seed=seed, V=V, T=T, dt=dt, speed_limit=sl, params=params, tap=tap, head=list(head),
                  ref_seconds_fwd=t1 - t0, ref_seconds_bwd=t2 - t1))


def gen_micro_rollouts(which):
    if "inv10" in which:     # example/inverse/micro.py defaults: 10 vehicles, dt 0.01; T=200
        micro_rollout("inv10", 10, 200, 0.01, 30.0, seed=1, params="default", tap="final_sq", record_steps=8)
    if "rand24" in which:    # random per-vehicle parameters, loss tapped every step
        micro_rollout("rand24", 24, 150, 1.0 / 30.0, 20.0, seed=4, params="random", tap="every_sum", record_steps=8)
    if "dense16" in which:   # C3 regime: 20 m spacing + U[0,10), v in [9,21], speed limit 30
        micro_rollout("dense16", 16, 400, 0.01, 30.0, seed=9, params="default", tap="final_sq", record_steps=4,
                      spacing=4.0, jitter=2.0, vlo=0.3, vhi=0.7)
    if "long" in which:      # T = 1000
        micro_rollout("long", 12, 1000, 0.01, 30.0, seed=12, params="default", tap="final_sq", record_steps=0)
    if "c3slice" in which:   # ONE lane of BASELINE config 3 at its full shape: 256 default vehicles x 1000 steps
        micro_rollout("c3slice", 256, 1000, 0.01, 30.0, seed=2028, params="default", tap="final_sq", record_steps=0)


# ----------------------------------------------------------------------------------------------
# G7: 3-lane hybrid network macro(0) -> micro(1) -> macro(2)  (example/inverse/hybrid.py:37-82)
# ----------------------------------------------------------------------------------------------

def gen_hybrid(name="hybrid3", N=10, T=500, dx=5.0, dt=0.01, um=30.0, seed=21):
    th.manual_seed(seed)
    np.random.seed(seed)
    L = N * dx
    bd_r = th.rand(4)
    bd_u = th.rand(4) * um
    r0 = (0.3 + 0.6 * th.rand(N)).requires_grad_(True)          # dense enough to emit vehicles
    u0 = (0.3 * um + 0.5 * um * th.rand(N)).requires_grad_(True)
    net = RoadNetwork(um)
    a = dMacroLane(0, L, um, dx)
    a.set_leftmost_cell(bd_r[0], bd_u[0])
    a.set_rightmost_cell(bd_r[1], bd_u[1])
    net.add_lane(a)
    a.set_state_vector_u(r0, u0)
    b = dMicroLane(1, L, um)
    net.add_lane(b)
    c = dMacroLane(2, L, um, dx)
    c.set_leftmost_cell(bd_r[2], bd_u[2])
    c.set_rightmost_cell(bd_r[3], bd_u[3])
    net.add_lane(c)
    net.connect_lane(0, 1)
    net.connect_lane(1, 2)
    net.macro_route = net.create_random_macro_route()
    events = []          # (step, kind, value): kind 0 = spawn on lane 1, 1 = hand-off to lane 2
    nveh = []
    t0 = time.time()
    for t in range(T):
        before, spawned = b.num_vehicle(), net.num_vehicle
        net.forward(dt, True)
        if net.num_vehicle > spawned:
            events.append((t, 0, float(b.curr_vehicle[0].speed)))
        if b.num_vehicle() < before + (net.num_vehicle - spawned):
            events.append((t, 1, float(c.curr_cell[0].state.q.r)))
        nveh.append(b.num_vehicle())
    rA, yA, uA = a.get_state_vector()
    rC, yC, uC = c.get_state_vector()
    pB, vB = b.get_state_vector()
    loss = (rC ** 2).sum() + (uC ** 2).sum() + (rA ** 2).sum() + (uA ** 2).sum()
    if len(pB):
        loss = loss + 1e-4 * (pB ** 2).sum() + (vB ** 2).sum()
    t1 = time.time()
    loss.backward()
    print("G7 %s: %d events, %d vehicles at the end, loss %.6f, fwd %.1fs bwd %.1fs" % (
        name, len(events), b.num_vehicle(), float(loss), t1 - t0, time.time() - t1))
    np.savez_compressed(
        os.path.join(OUT, "hybrid_%s.npz" % name),
        r0=r0.detach().numpy(), u0=u0.detach().numpy(), bd_r=bd_r.numpy(), bd_u=bd_u.numpy(),
        rA=rA.detach().numpy(), yA=yA.detach().numpy(), uA=uA.detach().numpy(),
        rC=rC.detach().numpy(), yC=yC.detach().numpy(), uC=uC.detach().numpy(),
        pB=pB.detach().numpy(), vB=vB.detach().numpy(), events=np.array(events, dtype=np.float64),
        nveh=np.array(nveh, dtype=np.int32), loss=np.float64(float(loss)),
        g_r0=r0.grad.numpy(), g_u0=u0.grad.numpy(),
        macro_next=np.array(sorted(net.macro_route.next_lane_dict.items()), dtype=np.int32),
        meta=meta(seed=seed, N=N, T=T, dx=dx, dt=dt, u_max=um))


# ----------------------------------------------------------------------------------------------
# G8: itscp environment (example/control/itscp): lane table, schedules, routes, reward and d reward / d action
# ----------------------------------------------------------------------------------------------

class SolverMargins:
    """Wraps ARZ.riemann_solve (calls it unchanged) and notes how close the run came to the solver's hard thresholds:
    the vacuum tests r_L < EPSILON / r_R < EPSILON and the equal-speed test |u_L - u_R| < EPSILON (_arz.py:225-257).
    A decision taken within float32 rounding of a threshold makes the run's gradient depend on the last bit of the
    torch build's float32 glue (SURVEY Note P), so goldens meant to pin a gradient log their minimum margin."""

    def __init__(self):
        self.orig = ARZ.riemann_solve
        self.n = 0
        self.min_equal = (np.inf, -1.0, -1.0)       # (margin, u_L, u_R)
        self.min_vacuum = np.inf
        self.n_equal = 0

    def __enter__(self):
        def wrapped(ql, qr, u_max):
            self.n += 1
            rl, rr = float(ql.q.r), float(qr.q.r)
            self.min_vacuum = min(self.min_vacuum, abs(rl - EPSILON))
            if rl >= EPSILON:
                self.min_vacuum = min(self.min_vacuum, abs(rr - EPSILON))
                if rr >= EPSILON:
                    d = abs(float(ql.u) - float(qr.u))
                    self.n_equal += d < EPSILON
                    m = abs(d - EPSILON)
                    if m < self.min_equal[0]:
                        self.min_equal = (m, float(ql.u), float(qr.u))
            return self.orig(ql, qr, u_max)
        ARZ.riemann_solve = staticmethod(wrapped)
        return self

    def __exit__(self, *a):
        ARZ.riemann_solve = staticmethod(self.orig)

    def as_meta(self):
        return dict(solver_calls=self.n, equal_speed_decisions=int(self.n_equal), min_equal_speed_margin=self.min_equal[0],
                    min_equal_speed_margin_at=[self.min_equal[1], self.min_equal[2]], min_vacuum_margin=self.min_vacuum)


def gen_itscp(name, mode, n_int, n_lane, lane_length, sim_len, sig_len, seed, action_kind, problem=1, differentiable=True,
              random_vehicles=False, action_override=None, draws_override=None, speed_limit=60.0):
    """action_override / draws_override (tools/probes/ref_replay_case.py): another action vector, and the admission draws of `micro` mode
    taken from a given stream instead of np.random -- the reference on a case a fuzz run found, to adjudicate between oracle and kernels.
    random_vehicles: every vehicle that enters the network takes the attributes of a MicroVehicle.random_micro_vehicle(speed_limit)
    (road/vehicle/micro_vehicle.py:75-121; the network code itself only ever builds default_micro_vehicle, conversion.py:51,
    road_network.py:582-591) drawn from a stream of its own (seeded; the run's other host randomness -- routes, admission draws -- is
    left as it would have been): the waiting vehicles of `micro` mode right after reset(), the flux capacitors' vehicles when they are
    added.  The fixture holds the six attributes per vehicle (veh_params: spawn order; waiting_params: per lane, list order).
    differentiable=False: an EVALUATION episode, env._simulate(action, False) under no_grad as Trainer.evaluate runs it
    (trainer.py:94-142): hard signal thresholds (_env.py:928-960), hard macro / micro boundaries (_simulator.py:116-137,
    264-276), hard is_static (_env.py:586-618).  The fixture then holds queues and reward only (no gradients)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_stubs"))
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp import problem as problems
    env = ItscpEnv()
    env.schedule_callback = getattr(problems, "problem_%d" % problem)
    env.render_eval = False
    for k, v in dict(num_intersection=n_int, lane_length=lane_length, num_lane=n_lane, render=False, policy_length=sim_len,
                     signal_length=sig_len, mode=mode, speed_limit=float(speed_limit), random_seed=seed).items():
        env.config[k] = v
    env.reset()
    sim = env.simulator
    veh_params, waiting_params = [], {}
    if random_vehicles:
        from road.vehicle.micro_vehicle import MicroVehicle
        veh_state = [np.random.RandomState(100000 + seed).get_state()]

        def randomize(nv):
            keep = np.random.get_state()
            np.random.set_state(veh_state[0])
            # (hybrid networks: drawn for 0.7 x the speed limit -- a random vehicle's target speed reaches 1.2 x its argument, and a
            # vehicle deposited into an ARZ cell above u_max ends the reference's own run in its CFL assert)
            rv = MicroVehicle.random_micro_vehicle(sim.speed_limit * float(random_vehicles))
            veh_state[0] = np.random.get_state()
            np.random.set_state(keep)
            for k in ("accel_max", "accel_pref", "target_speed", "min_space", "time_pref", "length"):
                setattr(nv, k, getattr(rv, k))
            return [float(getattr(nv, k)) for k in ("accel_max", "accel_pref", "target_speed", "min_space", "time_pref", "length")]
        if mode == "micro":
            for l, lst in sim.lane_waiting_micro_vehicle.items():
                waiting_params[int(l)] = [randomize(nv) for nv in lst]
        else:
            add_orig = sim.add_vehicle

            def add_logged(nv, nr):
                veh_params.append(randomize(nv))
                return add_orig(nv, nr)
            sim.add_vehicle = add_logged
    keys = list(env.lane.keys())
    nl = len(keys)
    T = env.num_timestep
    lane_tab = np.zeros((nl, 8), dtype=np.float64)      # sim id, is_macro, length, num_cell, cell_length, row, col, lane_id
    lane_str = []
    for i, k in enumerate(keys):
        sl = env.lane[k].sim_lane
        assert sl.id == i
        lane_tab[i] = (sl.id, float(sl.is_macro()), sl.length, getattr(sl, "num_cell", 0), getattr(sl, "cell_length", 0.0),
                       k.row, k.col, k.lane_id)
        lane_str.append("%s|%s|%d" % (k.loc, k.ploc, int(k.approaching)))
    edges = np.array([(a, b) for a in sim.lane for b in sim.lane[a].next_lane.keys()], dtype=np.int32)
    sched = np.array([env.schedule[k] for k in keys], dtype=np.float64)
    mroute = -np.ones((T, nl), dtype=np.int32)
    for t, r in enumerate(env.macro_route_schedule):
        for a, b in r.next_lane_dict.items():
            mroute[t, a] = b
    # routes drawn at spawn time, in call order
    spawn_routes = []
    orig = sim.create_random_route

    def logged(lane_id):
        r = orig(lane_id)
        spawn_routes.append(list(r.route))
        return r
    sim.create_random_route = logged
    A = env.action_size()
    rng = np.random.default_rng(seed)
    if action_kind == "half":
        a0 = np.full(A, 0.5, dtype=np.float32)
    else:
        a0 = rng.uniform(0.1, 0.9, A).astype(np.float32)
    if action_override is not None:
        a0 = np.asarray(action_override, dtype=np.float32)
        assert a0.shape == (A,)
    action = th.tensor(a0, requires_grad=True)
    t0 = time.time()
    env.queue_length.clear()
    # micro mode: source lanes admit a waiting vehicle when np.random.random() < inflow (_simulator.py:153-174): the draws
    # are host randomness, captured in call order like the routes
    rand_draws = []
    orig_random = np.random.random

    replay = None if draws_override is None else iter(np.asarray(draws_override, dtype=np.float64).tolist())

    def logged_random(*a, **kw):
        v = orig_random(*a, **kw)
        if replay is not None:
            v = np.full(np.shape(v), next(replay)) if np.shape(v) else next(replay)
        rand_draws.append(float(np.asarray(v).reshape(-1)[0]))
        return v
    waiting_routes = {int(l): [list(r.route) for r in rs] for l, rs in sim.lane_waiting_micro_route.items()}
    np.random.random = logged_random
    try:
        with SolverMargins() as margins:
            if differentiable:
                env._simulate(action, True)
            else:
                with th.no_grad():
                    env._simulate(action, False)
    finally:
        np.random.random = orig_random
    print("G8 %s: solver margins %s" % (name, margins.as_meta()))
    queue = np.array([[float(x) for x in env.queue_length[k]] for k in keys], dtype=np.float64)   # [lanes][T]
    reward = env._reward(action)
    t1 = time.time()
    if not differentiable:
        nveh = sim.num_vehicle
        print("G8 %s (evaluation episode): %d lanes, %d cells, T=%d, %d actions, %d vehicles spawned, reward %.6f, %.0fs" % (
            name, nl, int(lane_tab[:, 3].sum()), T, A, nveh, float(reward), t1 - t0))
        maxlen = max([len(r) for r in spawn_routes], default=1)
        sr = -np.ones((len(spawn_routes), maxlen), dtype=np.int32)
        for i, r in enumerate(spawn_routes):
            sr[i, :len(r)] = r
        np.savez_compressed(
            os.path.join(OUT, "itscp_%s.npz" % name),
            lane_tab=lane_tab, lane_str=np.array(lane_str), edges=edges, schedule=sched, macro_route=mroute, spawn_routes=sr,
            action=a0, reward=np.float64(float(reward)), queue=queue, rand_draws=np.array(rand_draws, dtype=np.float64),
            waiting_routes=np.array(json.dumps(waiting_routes)),
            veh_params=np.array(veh_params, dtype=np.float64).reshape(len(veh_params), 6), waiting_params=np.array(json.dumps(waiting_params)),
            meta=meta(seed=seed, mode=mode, num_intersection=n_int, num_lane=n_lane, lane_length=lane_length,
                      policy_length=sim_len, signal_length=sig_len, speed_limit=float(speed_limit), cell_length=5.0, simulation_frequency=30,
                      static_speed=0.2, vehicle_length=5.0, T=T, n_vehicle_spawned=nveh, problem=problem,
                      action_kind=action_kind, differentiable=False, **margins.as_meta(), ref_seconds_fwd=t1 - t0))
        return
    # split of the gradient by lane type (bisecting aid): reward restricted to macro / micro lanes
    parts = {}
    for tag, want_macro in (("macro", True), ("micro", False)):
        part = 0
        for k in keys:
            if env.lane[k].sim_lane.is_macro() == want_macro:
                for x in env.queue_length[k]:
                    part = part + (-1.0) * x
        if isinstance(part, th.Tensor) and part.requires_grad:
            parts[tag] = th.autograd.grad(part, action, retain_graph=True, allow_unused=True)[0]
            parts[tag] = np.zeros(len(a0), np.float32) if parts[tag] is None else parts[tag].numpy()
        else:
            parts[tag] = np.zeros(len(a0), np.float32)
    # gradient of the reward restricted to the first t0 steps (bisecting aid)
    cuts = [T // 4, T // 2, (3 * T) // 4] if os.environ.get("DHTS_FINE_CUTS") is None else [int(x) for x in os.environ["DHTS_FINE_CUTS"].split(",")]
    g_cut = []
    for t0_ in cuts:
        part = 0
        for k in keys:
            for x in env.queue_length[k][:t0_]:
                part = part + (-1.0) * x
        gc = th.autograd.grad(part, action, retain_graph=True, allow_unused=True)[0]
        g_cut.append(np.zeros(len(a0), np.float32) if gc is None else gc.numpy())
    # per-lane late contributions (bisecting aid): gradient of each macro lane's loss over the last quarter
    g_lane_late = {}
    if os.environ.get("DHTS_LANE_LATE"):
        want = [int(x) for x in os.environ["DHTS_LANE_LATE"].split(",")]
        for i, k in enumerate(keys):
            if i in want:
                part = 0
                for x in env.queue_length[k][(3 * T) // 4:]:
                    part = part + (-1.0) * x
                gc = th.autograd.grad(part, action, retain_graph=True, allow_unused=True)[0] if (isinstance(part, th.Tensor) and part.requires_grad) else None
                g_lane_late[i] = np.zeros(len(a0), np.float32) if gc is None else gc.numpy()
    reward.backward()
    t2 = time.time()
    nveh = sim.num_vehicle
    print("G8 %s: %d lanes, %d cells, T=%d, %d actions, %d vehicles spawned, reward %.6f, fwd %.0fs bwd %.0fs" % (
        name, nl, int(lane_tab[:, 3].sum()), T, A, nveh, float(reward), t1 - t0, t2 - t1))
    maxlen = max([len(r) for r in spawn_routes], default=1)
    sr = -np.ones((len(spawn_routes), maxlen), dtype=np.int32)
    for i, r in enumerate(spawn_routes):
        sr[i, :len(r)] = r
    np.savez_compressed(
        os.path.join(OUT, "itscp_%s.npz" % name),
        lane_tab=lane_tab, lane_str=np.array(lane_str), edges=edges, schedule=sched, macro_route=mroute, spawn_routes=sr,
        action=a0, reward=np.float64(float(reward)), g_action=action.grad.numpy(), queue=queue,
        g_action_macro_lanes=parts["macro"], g_action_micro_lanes=parts["micro"],
        g_action_cut_steps=np.array(cuts, dtype=np.int32), g_action_cut=np.array(g_cut, dtype=np.float32),
        rand_draws=np.array(rand_draws, dtype=np.float64),
        waiting_routes=np.array(json.dumps(waiting_routes)),
        veh_params=np.array(veh_params, dtype=np.float64).reshape(len(veh_params), 6), waiting_params=np.array(json.dumps(waiting_params)),
        g_lane_late_ids=np.array(sorted(g_lane_late), dtype=np.int32),
        g_lane_late=np.array([g_lane_late[i] for i in sorted(g_lane_late)], dtype=np.float32).reshape(len(g_lane_late), len(a0)),
        meta=meta(seed=seed, mode=mode, num_intersection=n_int, num_lane=n_lane, lane_length=lane_length,
                  policy_length=sim_len, signal_length=sig_len, speed_limit=float(speed_limit), cell_length=5.0, simulation_frequency=30,
                  static_speed=0.2, vehicle_length=5.0, T=T, n_vehicle_spawned=nveh, problem=problem,
                  action_kind=action_kind, **margins.as_meta(),
                  ref_seconds_fwd=t1 - t0, ref_seconds_bwd=t2 - t1))


# ----------------------------------------------------------------------------------------------
# G9: RoadNetwork.get_macro_state_of_micro_lane (road_network.py:207-297) on a micro -> micro -> micro chain
# ----------------------------------------------------------------------------------------------
def gen_macro_state_of_micro_lane(seed=3):
    from road.lane._micro_lane import MicroLane
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(6):
        sl = 30.0
        lengths = [float(x) for x in rng.uniform(20.0, 60.0, 3)]
        net = RoadNetwork(sl)
        for i in range(3):
            net.add_lane(MicroLane(i, lengths[i], sl))
        net.connect_lane(0, 1)
        net.connect_lane(1, 2)
        veh = []          # (lane, position, speed, route, route index)
        for lane_id, route, idx in ((0, [0, 1, 2], 0), (0, [0], 0), (1, [0, 1, 2], 1), (1, [1], 0), (2, [0, 1, 2], 2), (2, [2], 0)):
            pos_hi = lengths[lane_id]
            # one vehicle near the lane's far end (close to lane 1 for lane 0), one near its start (close to lane 1 for lane 2)
            pos = float(rng.uniform(0.8, 1.0) * pos_hi) if lane_id == 0 else float(rng.uniform(0.0, 0.2) * pos_hi) if lane_id == 2 \
                else float(rng.uniform(0.1, 0.9) * pos_hi)
            veh.append((lane_id, pos, float(rng.uniform(0.0, sl)), route, idx))
        veh.sort(key=lambda x: (x[0], x[1]))
        placed = []
        for lane_id, pos, spd, route, idx in veh:
            if any(l == lane_id and abs(p - pos) < 6.0 for l, p, *_ in placed):
                continue
            mv = MicroVehicle.default_micro_vehicle(sl)
            mv.position, mv.speed = pos, spd
            r = MicroRoute(route)
            for _k in range(idx):
                r.increment_curr_idx()
            net.add_vehicle(mv, r)
            placed.append((lane_id, pos, spd, route, idx))
        out = {}
        for flag in (True, False):
            d, s = net.get_macro_state_of_micro_lane(1, flag)
            out[flag] = (float(d), float(s))
        cases.append(dict(lengths=lengths, vehicles=placed, soft=out[True], hard=out[False]))
    np.savez_compressed(os.path.join(OUT, "macro_state_of_micro_lane.npz"), cases=np.array(json.dumps(cases)),
                        meta=meta(seed=seed, speed_limit=30.0))
    print("G9: %d cases, e.g. soft %s hard %s" % (len(cases), cases[0]["soft"], cases[0]["hard"]))


def main():
    global OUT
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="G1,G3,G4,G5,G6")
    ap.add_argument("--g4", default="c1,sanity,small,bench64,long")
    ap.add_argument("--g6", default="inv10,rand24,dense16,long")
    ap.add_argument("--g8", default="macro_small,macro,hybrid")
    args = ap.parse_args()
    only = set(args.only.split(","))
    os.makedirs(OUT, exist_ok=True)
    if "G1" in only:
        gen_riemann_kat()
    if "G1s" in only:
        gen_riemann_kat_stale()
    if "G3" in only:
        gen_macro_step()
    if "G4" in only:
        gen_macro_rollouts(set(args.g4.split(",")))
    if "G5" in only:
        gen_idm_kat()
    if "G5b" in only:
        gen_idm_kat_smallgap()
    if "G6" in only:
        gen_micro_rollouts(set(args.g6.split(",")))
    if "G7" in only:
        gen_hybrid()
        # other sizes, horizons, speed limits and initial states of the same three-lane network (round 6)
        gen_hybrid("hybrid3_b", N=16, T=400, seed=5)
        gen_hybrid("hybrid3_c", N=8, T=700, um=20.0, seed=9)
        gen_hybrid("hybrid3_d", N=12, T=600, dt=0.02, seed=33)
    if "G9" in only:
        gen_macro_state_of_micro_lane()
    if "G4x" in only or "G6x" in only:      # random straight lanes for tools/probes/lane_cases.py (untracked gpurun_in/, like G7x)
        OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_in")
        os.makedirs(OUT, exist_ok=True)
        rng = np.random.default_rng(4600)
        for k in range(48 if "G4x" in only else 0):
            N, T = int(rng.integers(3, 97)), int(rng.integers(10, 401))
            dx, um = float(rng.choice([2.5, 5.0, 10.0, 100.0])), float(rng.choice([15.0, 30.0]))
            dt = float(rng.choice([0.1, 0.2, 0.3]) * dx / um)             # (dt u_max / dx <= 0.3: inside the reference's CFL assert)
            try:
                macro_rollout("case_%d" % k, N, T, dx, dt, um, seed=int(rng.integers(1 << 16)), init=str(rng.choice(["uniform", "bench", "sanity"])),
                              tap=str(rng.choice(["final_sq", "every_sum"])), record_steps=int(min(T, 4)))
            except AssertionError as e:
                print("G4x case_%d: the reference asserts: %s" % (k, str(e)[:80]))
        rng = np.random.default_rng(4601)
        for k in range(48 if "G6x" in only else 0):
            V, T = int(rng.integers(1, 49)), int(rng.integers(10, 601))
            spacing = float(rng.choice([1.3, 2.0, 4.0]))                  # (in vehicle lengths; the jitter keeps add_vehicle's spacing assert)
            micro_rollout("case_%d" % k, V, T, float(rng.choice([0.01, 1.0 / 30.0, 0.05])), float(rng.choice([13.5, 20.0, 30.0])),
                          seed=int(rng.integers(1 << 16)), params=str(rng.choice(["default", "random"])), tap=str(rng.choice(["final_sq", "every_sum"])),
                          record_steps=int(min(T, 4)),
                          spacing=spacing, jitter=float(rng.choice([0.2, 0.5, 0.9])) * (spacing - 1.0))
    if "Gx_pick" in only:            # eighteen of the random cases above become committed fixtures (tests/golden/*_x<k>.npz): run after G4x,G6x,G7x
        import shutil
        src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_in")
        dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
        for k in (3, 5, 10, 22, 31, 40):
            shutil.copy(os.path.join(src, "macro_rollout_case_%d.npz" % k), os.path.join(dst, "macro_rollout_x%d.npz" % k))
        for k in (2, 9, 15, 23, 37, 44):
            shutil.copy(os.path.join(src, "micro_rollout_case_%d.npz" % k), os.path.join(dst, "micro_rollout_x%d.npz" % k))
        for k in ("0_1", "1_4", "2_11", "3_1", "4_0", "5_9"):
            shutil.copy(os.path.join(src, "hybrid_case_%s.npz" % k), os.path.join(dst, "hybrid_x%s.npz" % k))
    if "G7x" in only:                # 72 random cases of the three-lane network for tools/probes/three_lane_cases.py: not fixtures -- written to
        #                              the untracked gpurun_in/ (it travels to the GPU box with the snapshot), a few seconds each
        OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_in")
        os.makedirs(OUT, exist_ok=True)
        for w in range(6):
            rng = np.random.default_rng(300 + w)
            for k in range(12):
                N, T = int(rng.integers(4, 25)), int(rng.integers(200, 1001))
                um, dx = float(rng.choice([15.0, 20.0, 30.0, 35.0])), float(rng.choice([2.5, 5.0, 10.0]))
                dt = float(rng.choice([0.005, 0.01, 0.02, 0.04]))
                if dt * um / dx > 0.3:
                    dt = 0.01
                gen_hybrid("case_%d_%d" % (w, k), N=N, T=T, dx=dx, dt=dt, um=um, seed=int(rng.integers(1 << 16)))
    if "G8" in only:                 # run_itscp_macro.sh / run_itscp_hybrid.sh flag sets (+ a small macro case)
        which = set(args.g8.split(","))
        if "macro_small" in which:
            gen_itscp("macro_small", "macro", 1, 1, 10.0, 2, 1, seed=5, action_kind="rand")
        if "macro" in which:
            gen_itscp("macro", "macro", 1, 3, 30.0, 10, 2, seed=7, action_kind="rand")
        if "macro_2x2" in which:         # four intersections: lanes gated by a neighbouring intersection's signal
            gen_itscp("macro_2x2", "macro", 2, 2, 10.0, 4, 1, seed=17, action_kind="rand", problem=3)
        if "hybrid" in which:
            # bisecting aids recorded with this case: gradients of the reward restricted to its first t0 steps and
            # of the last-quarter loss of the lanes vehicles are deposited into (see tests/test_itscp_gpu.py)
            os.environ.setdefault("DHTS_FINE_CUTS", "150,300,450,480,510,540,570")
            os.environ.setdefault("DHTS_LANE_LATE", "16,54,82,116")
            gen_itscp(os.environ.get("DHTS_HYBRID_NAME", "hybrid"), "hybrid", 3, 1, 5.0, 20, 4, seed=9, action_kind="rand")
        if "hybrid_p2" in which:         # another inflow pattern (problem_2), seed and horizon (16 s, signal 4 s)
            os.environ["DHTS_FINE_CUTS"] = "120,240,360"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_p2", "hybrid", 3, 1, 5.0, 16, 4, seed=21, action_kind="rand", problem=2)
        if "hybrid_p3" in which:         # problem_3's inflow, signal length 3 s over 12 s (other phase grid), another seed
            os.environ["DHTS_FINE_CUTS"] = "90,180,270"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_p3", "hybrid", 3, 1, 5.0, 12, 3, seed=33, action_kind="rand", problem=3)
        if "hybrid_l10" in which:        # 10 m lanes (two cells each: deposits straddle cells, longer IDM lanes), problem_2
            os.environ["DHTS_FINE_CUTS"] = "100,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_l10", "hybrid", 3, 1, 10.0, 10, 2, seed=55, action_kind="rand", problem=2)
        # full-horizon pins of BASELINE config 4's episode (run_itscp_hybrid.sh: 3 x 3, 1 lane, 5 m, 20 s, signal 4 s = 600
        # steps): action 0.5 (every signal sigmoid at its steepest point) and other seeds / inflow patterns;
        # meta carries the run's minimum distance to the solver's equal-speed threshold
        for nm, sd, kind, prob in (("hybrid_half", 9, "half", 1), ("hybrid_s2", 41, "rand", 1), ("hybrid_s3", 77, "rand", 1),
                                   ("hybrid_p2_600", 63, "rand", 2), ("hybrid_half_p3", 15, "half", 3)):
            if nm in which:
                os.environ["DHTS_FINE_CUTS"] = "150,300,450,540"
                os.environ.pop("DHTS_LANE_LATE", None)
                gen_itscp(nm, "hybrid", 3, 1, 5.0, 20, 4, seed=sd, action_kind=kind, problem=prob)
        if "micro" in which:             # run_itscp_micro.sh's flags (plain autodiff MicroLane everywhere, stochastic source lanes)
            os.environ["DHTS_FINE_CUTS"] = "150"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro", "micro", 1, 3, 30.0, 10, 2, seed=13, action_kind="rand")
        # further differentiable `micro`-mode runs (end of round 5: more pins of the float32 tensor ladder): two lanes per approach over
        # problem_2's inflows, 6 s; one lane per approach with 10 m lanes (vehicles leave their lane every few steps), 1 s signals
        if "micro_p2" in which:
            os.environ["DHTS_FINE_CUTS"] = "90"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro_p2", "micro", 1, 2, 30.0, 6, 2, seed=23, action_kind="rand", problem=2)
        if "micro_l10" in which:
            os.environ["DHTS_FINE_CUTS"] = "90"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro_l10", "micro", 1, 1, 10.0, 6, 1, seed=31, action_kind="rand", problem=3)
        if "micro_small" in which:       # one lane per approach, 4 s
            os.environ["DHTS_FINE_CUTS"] = "60"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro_small", "micro", 1, 1, 30.0, 4, 2, seed=3, action_kind="rand")
        if "macro_half" in which:        # run_itscp_macro.sh's episode at action 0.5
            gen_itscp("macro_half", "macro", 1, 3, 30.0, 10, 2, seed=8, action_kind="half")
        if "macro_long" in which:        # the same network for 15 s: 236 cells x 450 steps = 106 200 loss samples > the
            gen_itscp("macro_long", "macro", 1, 3, 30.0, 15, 3, seed=19, action_kind="rand", problem=2)   # RunningMean window of 100 000
        # evaluation episodes (differentiable = False): the macro network, the 240-step hybrid episode, BASELINE config 4's
        # full 600-step hybrid episode, and the hybrid network over problem_2's inflows
        if "eval_macro" in which:
            gen_itscp("eval_macro", "macro", 1, 3, 30.0, 10, 2, seed=7, action_kind="rand", differentiable=False)
        if "eval_macro_2x2" in which:
            gen_itscp("eval_macro_2x2", "macro", 2, 2, 10.0, 4, 1, seed=17, action_kind="rand", problem=3, differentiable=False)
        if "eval_hybrid_short" in which:
            gen_itscp("eval_hybrid_short", "hybrid", 3, 1, 5.0, 8, 2, seed=9, action_kind="rand", differentiable=False)
        if "eval_hybrid" in which:
            gen_itscp("eval_hybrid", "hybrid", 3, 1, 5.0, 20, 4, seed=9, action_kind="rand", differentiable=False)
        if "eval_hybrid_p2" in which:
            gen_itscp("eval_hybrid_p2", "hybrid", 3, 1, 5.0, 16, 4, seed=21, action_kind="rand", problem=2, differentiable=False)
        # 4 x 4 intersections: the interior 2 x 2 are micro -- 64 IDM lanes, 8 macro lanes feeding them (the fused kernels'
        # capacity since round 3; run_itscp_hybrid.sh with --n_intersection=4), 8 s
        if "hybrid_4x4" in which:
            os.environ["DHTS_FINE_CUTS"] = "120,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_4x4", "hybrid", 4, 1, 5.0, 8, 2, seed=29, action_kind="rand", problem=2)
        if "hybrid_n2" in which:         # two lanes per approach: 28 micro lanes at the centre intersection, 12 macro lanes feeding them
            os.environ["DHTS_FINE_CUTS"] = "120,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_n2", "hybrid", 3, 2, 5.0, 8, 2, seed=37, action_kind="rand", problem=1)
        # evaluation episodes in `micro` mode (round 5): Python floats all the way in the reference (no tensors without gradients)
        if "eval_micro_small" in which:
            gen_itscp("eval_micro_small", "micro", 1, 1, 30.0, 4, 2, seed=3, action_kind="rand", differentiable=False)
        if "eval_micro" in which:
            gen_itscp("eval_micro", "micro", 1, 3, 30.0, 10, 2, seed=13, action_kind="rand", differentiable=False)
        if "eval_hybrid_4x4" in which:
            gen_itscp("eval_hybrid_4x4", "hybrid", 4, 1, 5.0, 8, 2, seed=29, action_kind="rand", problem=2, differentiable=False)
        # networks ABOVE the fused kernels' per-workgroup limits (round 5: pins of the stepwise batched path and of the
        # oracle at sizes only the oracle used to judge): 360 macro lanes / 1 692 cells; 30 m lanes in hybrid mode
        # (~1 000 cells + 144 lanes); 5 x 5 intersections (144 IDM lanes); micro mode with 112 IDM lanes
        if "macro_3x3x3" in which:
            os.environ["DHTS_FINE_CUTS"] = "60,90"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("macro_3x3x3", "macro", 3, 3, 30.0, 4, 2, seed=61, action_kind="rand", problem=2)
        if "hybrid_l30" in which:
            os.environ["DHTS_FINE_CUTS"] = "120,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_l30", "hybrid", 3, 1, 30.0, 8, 2, seed=71, action_kind="rand", problem=2)
        if "hybrid_n2l30" in which:      # two lanes per approach, 30 m lanes: 252 lanes + ~1 300 cells (beyond one workgroup)
            os.environ["DHTS_FINE_CUTS"] = "120,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_n2l30", "hybrid", 3, 2, 30.0, 8, 2, seed=79, action_kind="rand", problem=1)
        if "hybrid_5x5" in which:
            os.environ["DHTS_FINE_CUTS"] = "120,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_5x5", "hybrid", 5, 1, 5.0, 8, 2, seed=73, action_kind="rand", problem=1)
        if "micro_2x2" in which:
            os.environ["DHTS_FINE_CUTS"] = "60"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro_2x2", "micro", 2, 2, 10.0, 4, 1, seed=83, action_kind="rand", problem=3)
        # evaluation episodes of the networks above the fused kernels' limits (end of round 5: pins of the stepwise path's evaluation
        # instantiation at those sizes)
        if "eval_macro_3x3x3" in which:
            gen_itscp("eval_macro_3x3x3", "macro", 3, 3, 30.0, 4, 2, seed=61, action_kind="rand", problem=2, differentiable=False)
        if "eval_hybrid_n2l30" in which:
            gen_itscp("eval_hybrid_n2l30", "hybrid", 3, 2, 30.0, 8, 2, seed=79, action_kind="rand", problem=1, differentiable=False)
        if "eval_hybrid_5x5" in which:
            gen_itscp("eval_hybrid_5x5", "hybrid", 5, 1, 5.0, 8, 2, seed=73, action_kind="rand", problem=1, differentiable=False)
        if "eval_micro_2x2" in which:
            gen_itscp("eval_micro_2x2", "micro", 2, 2, 10.0, 4, 1, seed=83, action_kind="rand", problem=3, differentiable=False)
        # vehicles with their own IDM attributes (round 6: the per-vehicle parameter table of the network kernels): the hybrid network over
        # problem_2's inflows for 16 s, and `micro` mode for 4 s / on a 2 x 2 x 2 grid
        if "hybrid_rv" in which:
            os.environ["DHTS_FINE_CUTS"] = "120,240,360"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_rv", "hybrid", 3, 1, 5.0, 16, 4, seed=121, action_kind="rand", problem=2, random_vehicles=0.7)
        # 12 s, other seeds: fixtures whose WHOLE gradient is well conditioned (hybrid_rv's moves by percents under a one-ulp change of
        # the action: tests/test_oracle_golden.py shows it and pins that run on the reward's first 360 steps)
        for nm, sd in (("hybrid_rv_b", 122), ("hybrid_rv_d", 124)):
            if nm in which:
                os.environ["DHTS_FINE_CUTS"] = "120,240,300"
                os.environ.pop("DHTS_LANE_LATE", None)
                gen_itscp(nm, "hybrid", 3, 1, 5.0, 12, 4, seed=sd, action_kind="rand", problem=2, random_vehicles=0.7)
        if "hybrid_rv_l10" in which:
            os.environ["DHTS_FINE_CUTS"] = "100,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_rv_l10", "hybrid", 3, 1, 10.0, 10, 2, seed=155, action_kind="rand", problem=2, random_vehicles=0.7)
        if "micro_rv" in which:
            os.environ["DHTS_FINE_CUTS"] = "60"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro_rv", "micro", 1, 1, 30.0, 4, 2, seed=103, action_kind="rand", random_vehicles=1.0)
        if "micro_rv_2x2" in which:
            os.environ["DHTS_FINE_CUTS"] = "60"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro_rv_2x2", "micro", 2, 2, 10.0, 4, 1, seed=183, action_kind="rand", problem=3, random_vehicles=1.0)
        if "micro_rv_l10" in which:      # 10 m lanes: vehicles leave their lane every few steps, each with its own attributes
            os.environ["DHTS_FINE_CUTS"] = "90"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("micro_rv_l10", "micro", 1, 1, 10.0, 6, 1, seed=231, action_kind="rand", problem=3, random_vehicles=1.0)
        if "hybrid_rv_n2" in which:      # two lanes per approach: 28 IDM lanes, 12 spawning lanes
            os.environ["DHTS_FINE_CUTS"] = "120,200"
            os.environ.pop("DHTS_LANE_LATE", None)
            gen_itscp("hybrid_rv_n2", "hybrid", 3, 2, 5.0, 8, 2, seed=137, action_kind="rand", problem=1, random_vehicles=0.7)
        if "eval_hybrid_rv" in which:
            gen_itscp("eval_hybrid_rv", "hybrid", 3, 1, 5.0, 16, 4, seed=121, action_kind="rand", problem=2, differentiable=False, random_vehicles=0.7)
        # congested `micro` mode episodes (8 s, 60 m lanes, ~110 vehicles): followers close in below POSITION_DELTA_EPS and collide
        # ("Set deltas to 0"), where autograd differentiates the forward's clamps -- constants -- and the gradient stays finite
        # (found by tools/probes/fuzz_env.py: dIDM's formulas at the un-clamped gap divide by zero there)
        for nm, sd, pb in (("micro_jam_a", 1, 2), ("micro_jam_b", 14, 3), ("micro_jam_c", 25, 2)):
            if nm in which:
                os.environ.pop("DHTS_FINE_CUTS", None)
                os.environ.pop("DHTS_LANE_LATE", None)
                gen_itscp(nm, "micro", 1, 3, 60.0, 8, 1, seed=sd, action_kind="rand", problem=pb)
        # seven episodes of tools/probes/ref_sweep.py's random shapes: speed limits other than 60 m/s, episodes with more than 128 vehicles
        # (beyond the fused kernels: the stepwise path), a 252-lane `micro` grid
        for nm, a_ in (("sweep_a", ("macro", 2, 2, 20.0, 6, 1, 57327, 2, False, 30.0)), ("sweep_b", ("macro", 1, 3, 10.0, 4, 2, 19269, 1, True, 30.0)),
                       ("sweep_c", ("micro", 2, 2, 20.0, 8, 4, 24407, 2, True, 30.0)), ("sweep_d", ("micro", 2, 3, 10.0, 10, 2, 13207, 2, False, 30.0)),
                       ("sweep_e", ("micro", 3, 2, 30.0, 6, 2, 45800, 2, True, 30.0)), ("sweep_f", ("hybrid", 3, 1, 10.0, 16, 2, 43887, 3, False, 45.0)),
                       ("sweep_g", ("hybrid", 3, 1, 10.0, 16, 4, 1323, 1, True, 45.0))):
            if nm in which:
                os.environ.pop("DHTS_FINE_CUTS", None)
                os.environ.pop("DHTS_LANE_LATE", None)
                gen_itscp(nm, a_[0], a_[1], a_[2], a_[3], a_[4], a_[5], seed=a_[6], action_kind="rand", problem=a_[7], differentiable=a_[8], speed_limit=a_[9])
        if "hybrid_short" in which:      # 8 s: enough for the first vehicles to cross the interior intersection
            gen_itscp("hybrid_short", "hybrid", 3, 1, 5.0, 8, 2, seed=9, action_kind="rand")


if __name__ == "__main__":
    main()
