"""Parity of the HIP path (through the C ABI, libdhts.so) against the CPU oracle and the golden vectors.
Needs a real MI355X:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

from util import TOL_GRAD, TOL_STATE, grad_report, meta_of, rel_elem, rel_max, ulp_diff

pytestmark = pytest.mark.gpu


def T_(x, dev, dtype=None, grad=False):
    import torch
    t = torch.tensor(np.ascontiguousarray(x), device=dev, dtype=dtype)
    if grad:
        t.requires_grad_(True)
    return t


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def tape_to_dqs(tape, T, L, N, planes=3):
    """[T][L][planes][Np][4] -> the reference's dqs [T][L][N][planes][2][2]."""
    Np = (N + 63) // 64 * 64
    t = tape.cpu().numpy().reshape(T, L, planes, Np, 4)[:, :, :, :N, :]
    return np.ascontiguousarray(t.transpose(0, 1, 3, 2, 4)).reshape(T, L, N, planes, 2, 2)


def rollout_tape_to_dqs(desc, tape, T):
    """The rollout tape -> dqs [T][L][N][3][2][2] through dhts_macro_tape_expand (the reverse sweep's own code: trivial
    interfaces recomputed from the stored left-cell state, the others read from the compacted exceptions)."""
    from dhts import ops
    return tape_to_dqs(ops.macro_tape_expand(desc, T, tape), T, desc.n_lanes, desc.n_cells)


def dqs_to_tape(dqs, planes=3):
    T, L, N = dqs.shape[:3]
    Np = (N + 63) // 64 * 64
    out = np.zeros((T, L, planes, Np, 4), np.float32)
    out[:, :, :, :N, :] = dqs.reshape(T, L, N, planes, 4).transpose(0, 1, 3, 2, 4)
    return out


# =================================================================================================================
# known-answer vectors straight through the device math
# =================================================================================================================
@pytest.mark.parametrize("variant", [0, 1])
def test_device_interface_solver_vs_reference_kat(cuda, golden_dir, variant):
    """G1/G2 on the device: every (branch, case) pair of the Riemann solver, Q_0, Jacobians, flux Jacobian and
    the 2x2 products, for the production arithmetic (0) and the reference-order IEEE build (1)."""
    import torch
    from dhts import ops
    g = load(golden_dir, "riemann_kat.npz")
    out = ops.arz_interface_batch(T_(g["inp"], cuda), variant=variant)
    case = out["case"].cpu().numpy()
    assert np.array_equal(case, g["case"])                     # all 2076 branch decisions
    assert set(zip(g["branch"].tolist(), case.tolist())) == {(1, 0), (2, 0), (2, 2), (3, 0), (4, 0), (4, 1), (5, 0),
                                                             (5, 1), (5, 2), (6, 0), (6, 2)}
    q0 = out["q0"].cpu().numpy()
    tol = 1e-13 if variant == 1 else 2e-10                     # double results: a few ulps (x cancellation in u = y / r + u_eq)
    assert np.max(np.abs(q0 - g["q0"]) / np.maximum(np.abs(g["q0"]), 1e-3)) <= tol
    flux_ref = np.stack([g["q0"][:, 0] * g["q0"][:, 2], g["q0"][:, 1] * g["q0"][:, 2]], 1)
    assert np.max(np.abs(out["flux"].cpu().numpy() - flux_ref) / np.maximum(np.abs(flux_ref), 1e-3)) <= 10 * tol
    for key in ("dL", "dR", "fp"):
        # float32 entries: almost all bit-exact; an entry that is a near-cancellation of two O(1) terms may differ
        # in its last double ulps, so the bound is one float32 ulp of the matrix' largest entry
        got, ref = out[key].cpu().numpy(), g[key]
        scale = np.abs(ref).reshape(-1, 4).max(1).reshape(-1, 1, 1)
        assert np.all(np.abs(got - ref) <= 1.2e-7 * np.maximum(scale, 1e-30)), key
        assert np.mean(got != ref) <= (2e-3 if variant == 1 else 2e-2), key
    # products as np.matmul forms them (acc = a0*b0; acc = fma(a1, b1, acc)) from the reference's own factors
    def mm(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        o = np.empty_like(a)
        for i in range(2):
            for j in range(2):
                p0 = (a[:, i, 0] * b[:, 0, j]).astype(np.float32).astype(np.float64)
                o[:, i, j] = p0 + a[:, i, 1] * b[:, 1, j]
        return o.astype(np.float32)
    same = np.all((out["dL"].cpu().numpy() == g["dL"]) & (out["dR"].cpu().numpy() == g["dR"]) &
                  (out["fp"].cpu().numpy() == g["fp"]), axis=(1, 2))
    assert same.mean() >= (0.99 if variant == 1 else 0.95)
    A_ref, B_ref = mm(g["fp"], g["dL"]), mm(g["fp"], g["dR"])
    assert np.array_equal(out["A"].cpu().numpy()[same], A_ref[same])
    assert np.array_equal(out["B"].cpu().numpy()[same], B_ref[same])
    # the CFL assert of _macro_lane.py:141-146 evaluated on the reference's own wave speeds (dt = 0.01, dx = 5)
    sp = np.maximum(np.abs(g["speed"]), 1e-5)
    cfl_ref = ~((0.01 < 5.0 / sp[:, 0]) & (0.01 < 5.0 / sp[:, 1]))
    assert np.array_equal(out["cfl_bad"].cpu().numpy(), cfl_ref) and cfl_ref.any() and not cfl_ref.all()
    # speed0, speed1 as ARZ.riemann_solve returns them (_arz.py:316-332): reference-order arithmetic in both variants; the
    # reference's r ** (gamma - 1) is one libm pow where the device takes 1 / sqrt(r) (two roundings): a few double ulps
    sp_dev = out["speed"].cpu().numpy()
    assert np.max(np.abs(sp_dev - g["speed"]) / np.maximum(np.abs(g["speed"]), 1e-3)) <= 1e-13
    assert np.mean(sp_dev == g["speed"]) >= 0.9


@pytest.mark.parametrize("variant", [0, 1])
def test_device_interface_solver_on_deposited_cells(cuda, golden_dir, variant):
    """Cells rewritten by Conversion.micro_to_macro carry u, y of the deposited vehicle but the u_eq of the density
    before the deposit (conversion.py:158-166): the solver must read each of r, y, u, u_eq where the reference does."""
    from dhts import ops
    g = load(golden_dir, "riemann_kat_stale.npz")
    out = ops.arz_interface_batch(T_(g["inp"], cuda), variant=variant)
    assert np.array_equal(out["case"].cpu().numpy(), g["case"])
    tol = 1e-13 if variant == 1 else 2e-10
    q0 = out["q0"].cpu().numpy()
    assert np.max(np.abs(q0 - g["q0"]) / np.maximum(np.abs(g["q0"]), 1e-3)) <= tol
    for key in ("dL", "dR", "fp"):
        got, ref = out[key].cpu().numpy(), g[key]
        scale = np.abs(ref).reshape(-1, 4).max(1).reshape(-1, 1, 1)
        assert np.all(np.abs(got - ref) <= 1.2e-7 * np.maximum(scale, 1e-30)), key


@pytest.mark.parametrize("variant", [0, 1])
def test_device_idm_vs_reference_kat(cuda, golden_dir, variant):
    """G5 on the device: acceleration with both clips, Euler step, dEgo / dLeading, for the production arithmetic (0)
    and the reference-order IEEE version (1)."""
    from dhts import ops
    g = load(golden_dir, "idm_kat.npz")
    inp = g["inp"]
    out = ops.idm_batch(T_(inp, cuda), variant=variant)
    v, dt = inp[:, 2], inp[:, 8]
    assert np.array_equal(out["clipped_acc"].cpu().numpy(), g["flags"][:, 0].astype(bool))
    assert np.array_equal(out["clipped_spacing"].cpu().numpy(), g["flags"][:, 1].astype(bool))
    assert np.max(np.abs(out["acc"].cpu().numpy() - g["acc"]) / np.maximum(np.abs(g["acc"]), 1e-3)) <= 1e-12
    assert np.max(np.abs(out["sstar"].cpu().numpy() - g["sstar"])) <= 1e-12
    nv_ref = (v + dt * g["acc"]).astype(np.float32)
    assert ulp_diff(out["next_v"].cpu().numpy().astype(np.float32), nv_ref).max() <= 1
    for key in ("dEgo", "dLeading"):
        got, ref = out[key].cpu().numpy(), g[key]
        assert rel_max(got, ref) <= 1e-6
        d = ulp_diff(got, ref)
        assert d.max() <= 2 and np.mean(d > 0) <= 0.01, key
    assert not out["collided"].any()


# =================================================================================================================
# macro
# =================================================================================================================
@pytest.mark.parametrize("name", ["rand64", "sanity100", "vacuum9", "single1", "jam33"])
def test_macro_step_vs_golden_and_oracle(cuda, oracle, golden_dir, name):
    """dhts_macro_step_fwd / _bwd against the reference's own single-step outputs (G3)."""
    import torch
    from dhts import ops
    g = load(golden_dir, "macro_step.npz")
    c = meta_of(g)["configs"][name]
    st = g[name + "_state"]
    N = c["N"]
    desc = ops.macro_desc(1, N, c["dt"], c["dx"], c["u_max"])
    planes = [T_(st[k, 1:-1][None], cuda) for k in range(4)]
    ghost = T_(np.stack([st[:, 0], st[:, -1]])[None], cuda)           # [1][2][4]
    tape = torch.zeros(ops.macro_step_tape_numel(desc), device=cuda)
    err = ops.new_error_record(cuda)
    nr, ny, nu, nq = ops.macro_step_fwd(desc, *planes, ghost, tape=tape, err=err)
    assert ops.raise_on_fault(err) == 0
    dqs = tape_to_dqs(tape, 1, 1, N)[0, 0]
    # the fused rollout entry point (interface tape) gives the same step and the same blocks
    tape_i = torch.zeros(ops.macro_tape_numel(desc, 1), device=cuda)
    out_i = ops.macro_rollout_fwd(desc, 1, *planes, ghost, tape=tape_i)
    assert all(torch.equal(a, b) for a, b in zip(out_i, (nr, ny, nu, nq)))
    assert np.array_equal(rollout_tape_to_dqs(desc, tape_i, 1)[0, 0], dqs)
    o = oracle.macro_step(st, c["dt"], c["dx"], c["u_max"])
    # device double arithmetic follows the oracle's operation order; only pow() vs sqrt()/mul differs
    assert ulp_diff(nr.cpu().numpy()[0], o["nr"]).max() <= 1
    assert ulp_diff(ny.cpu().numpy()[0], o["ny"]).max() <= 1
    # u = y / r + u_eq cancels, so compare it norm-relative against the oracle's ...
    assert rel_max(nu.cpu().numpy()[0], o["nu"]) <= 1e-6
    assert rel_max(nq.cpu().numpy()[0], o["nueq"]) <= 1e-6
    # ... and exactly (<= 1 ulp: the device has no correctly rounded float32 division by default) against
    # the oracle's float32 glue evaluated on the device's own (r, y)
    glue = np.array([oracle.arz_from_r_y(a, b, c["u_max"]) for a, b in zip(nr.cpu().numpy()[0], ny.cpu().numpy()[0])])
    assert ulp_diff(nu.cpu().numpy()[0], glue[:, 0]).max() <= 1
    assert ulp_diff(nq.cpu().numpy()[0], glue[:, 1]).max() <= 1
    assert rel_max(dqs, o["dqs"]) <= 1e-6
    assert np.mean(dqs != o["dqs"]) <= 0.02
    # and against the reference's own numbers
    assert rel_max(nr.cpu().numpy()[0], g[name + "_nr"]) <= 2e-7
    assert rel_max(ny.cpu().numpy()[0], g[name + "_ny"]) <= 2e-7
    assert rel_max(dqs, g[name + "_dqs"]) <= 1e-6
    # backward on the reference's tape and cotangents
    tape_ref = T_(dqs_to_tape(g[name + "_dqs"][None, None]).reshape(-1), cuda)
    g_r0, g_y0, g_ghost = ops.macro_step_bwd(desc, tape_ref, T_(g[name + "_g_nr"][None], cuda), T_(g[name + "_g_ny"][None], cuda))
    # the rollout reverse sweep on the interface tape of the same step agrees with the operator on its own tape
    b_i = ops.macro_rollout_bwd(desc, 1, tape_i, T_(g[name + "_g_nr"][None], cuda), T_(g[name + "_g_ny"][None], cuda))
    b_s = ops.macro_step_bwd(desc, tape, T_(g[name + "_g_nr"][None], cuda), T_(g[name + "_g_ny"][None], cuda))
    assert all(torch.equal(a, b) for a, b in zip(b_i, b_s))
    ref_r, ref_y = g[name + "_g_r"], g[name + "_g_y"]
    assert np.array_equal(g_r0.cpu().numpy()[0], ref_r[1:-1]) and np.array_equal(g_y0.cpu().numpy()[0], ref_y[1:-1])
    gg = g_ghost.cpu().numpy()[0]
    assert np.allclose(gg[0], [ref_r[0], ref_y[0]], rtol=1e-6, atol=1e-7) and np.allclose(gg[1], [ref_r[-1], ref_y[-1]], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["small", "c1", "sanity", "bench64", "long", "x3", "x5", "x10", "x22", "x31", "x40"])   # x<k>: random shapes (gen_goldens --only G4x,Gx_pick)
def test_macro_rollout_vs_golden(cuda, golden_dir, name):
    """Autograd-level drop-in (dhts.macro_rollout) against the reference's rollouts and gradients (G4)."""
    import torch
    import dhts
    g = load(golden_dir, "macro_rollout_%s.npz" % name)
    m = meta_of(g)
    r0, u0 = T_(g["r0"][None], cuda, grad=True), T_(g["u0"][None], cuda, grad=True)
    gr, gu = T_(g["ghost_r"][None], cuda, grad=True), T_(g["ghost_u"][None], cuda, grad=True)
    every = m["tap"] == "every_sum"
    out = dhts.macro_rollout(r0, u0, gr, gu, m["T"], m["dt"], m["dx"], m["u_max"], want_hist=True)
    rT, yT, uT, _, hist = out
    if every:
        loss = hist.sum()
    else:
        loss = (rT ** 2).sum() + (uT ** 2).sum()
    loss.backward()
    assert rel_max(rT.detach().cpu().numpy()[0], g["rT"]) <= TOL_STATE
    assert rel_max(yT.detach().cpu().numpy()[0], g["yT"]) <= TOL_STATE
    assert rel_max(uT.detach().cpu().numpy()[0], g["uT"]) <= TOL_STATE
    # element-wise as well (floor 1e-6 max|ref| under the division): "state <= 1e-5 relative" entry by entry
    e_elem = max(rel_elem(rT.detach().cpu().numpy()[0], g["rT"]), rel_elem(uT.detach().cpu().numpy()[0], g["uT"]))
    print("G4 %s (kernels): element-wise state error %.2e" % (name, e_elem))
    assert e_elem <= TOL_STATE
    h = hist.detach().cpu().numpy()
    for t in range(len(g["steps_r"])):
        assert rel_max(h[t, 0, 0], g["steps_r"][t]) <= TOL_STATE
        assert rel_max(h[t, 0, 1], g["steps_y"][t]) <= TOL_STATE
        assert rel_max(h[t, 0, 2], g["steps_u"][t]) <= TOL_STATE
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    assert grad_report("G4 %s (kernels) d loss / d r0" % name, r0.grad.cpu().numpy()[0], g["g_r0"]) <= TOL_GRAD
    assert grad_report("G4 %s (kernels) d loss / d u0" % name, u0.grad.cpu().numpy()[0], g["g_u0"]) <= TOL_GRAD
    assert rel_max(gr.grad.cpu().numpy()[0], g["g_ghost_r"]) <= TOL_GRAD
    assert rel_max(gu.grad.cpu().numpy()[0], g["g_ghost_u"]) <= TOL_GRAD


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 100, 127, 128, 129, 512, 1000, 1030, 2048, 2500])
def test_macro_rollout_vs_oracle_sizes(cuda, oracle, N):
    """Lane lengths around the 64-cell pass boundaries, several lanes, against the oracle."""
    import torch
    import dhts
    rng = np.random.default_rng(100 + N)
    L, T, dt, dx, um = 5, 25, 0.01, 5.0, 30.0
    r0 = rng.uniform(0.0, 1.0, (L, N)).astype(np.float32)
    u0 = rng.uniform(0.0, um, (L, N)).astype(np.float32)
    if N > 8:
        r0[0, 3:6] = 0.0          # vacuum cells
        r0[1, N // 2] = 1e-6
    gr = rng.uniform(0.0, 1.0, (L, 2)).astype(np.float32)
    gu = rng.uniform(0.0, um, (L, 2)).astype(np.float32)
    f = oracle.macro_rollout_fwd(r0, u0, gr, gu, T, dt, dx, um)
    assert f["rc"] == 0
    w_r = rng.standard_normal((L, N)).astype(np.float32)
    w_u = rng.standard_normal((L, N)).astype(np.float32)
    b = oracle.macro_rollout_bwd(f, g_rT=w_r, g_uT=w_u)
    tr0, tu0 = T_(r0, cuda, grad=True), T_(u0, cuda, grad=True)
    tgr, tgu = T_(gr, cuda, grad=True), T_(gu, cuda, grad=True)
    rT, yT, uT, _ = dhts.macro_rollout(tr0, tu0, tgr, tgu, T, dt, dx, um)
    ((rT * T_(w_r, cuda)).sum() + (uT * T_(w_u, cuda)).sum()).backward()
    assert rel_max(rT.detach().cpu().numpy(), f["rT"]) <= TOL_STATE
    assert rel_max(yT.detach().cpu().numpy(), f["yT"]) <= TOL_STATE
    assert rel_max(uT.detach().cpu().numpy(), f["uT"]) <= TOL_STATE
    assert rel_max(tr0.grad.cpu().numpy(), b["g_r0"]) <= TOL_GRAD
    assert rel_max(tu0.grad.cpu().numpy(), b["g_u0"]) <= TOL_GRAD
    assert rel_max(tgr.grad.cpu().numpy(), b["g_ghost_r"]) <= TOL_GRAD
    assert rel_max(tgu.grad.cpu().numpy(), b["g_ghost_u"]) <= TOL_GRAD


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("waves", [1, 2, 3, 5, 8, 16])
def test_macro_forward_is_independent_of_wave_split(cuda, waves, variant):
    """The forward kernels split a lane over 1..16 wavefronts, and the two-phase kernel (variant 0) solves an interface
    either in place or from its queue depending on the split; state, history and tape must not depend on any of it:
    bitwise within a kernel.  Between the two kernels the interface solves are bitwise the same too; only the float32 glue
    (u, u_eq from r, y) differs -- IEEE float32 sqrt / division in the one-phase kernel, roundings of the double-precision
    square roots in the two-phase kernel, equal except with probability ~1e-6 per operation -- so across kernels the
    comparison is at 1e-6 instead of bitwise."""
    import torch
    from dhts import _lib, ops
    if variant == 1 and waves > 8:
        pytest.skip("the one-phase kernel takes at most 8 wavefronts per lane")
    rng = np.random.default_rng(77)
    L, N, T, dt, dx, um = 6, 517, 40, 0.01, 5.0, 30.0
    r0 = rng.uniform(0.0, 1.0, (L, N)).astype(np.float32)
    r0[2, 100:104] = 0.0           # vacuum cells
    r0[3, 0] = 1e-6
    r0[4, 200] = np.float32(1e-5)  # exactly float32(eps): the glue divides by r, the solver clamps to the double eps
    r = T_(r0, cuda)
    u = T_(rng.uniform(0.0, um, (L, N)).astype(np.float32), cuda)
    y, q = ops.macro_state_from_ru(r, u, um)
    gr = T_(rng.uniform(0.0, 1.0, (L, 2)).astype(np.float32), cuda)
    gu = T_(rng.uniform(0.0, um, (L, 2)).astype(np.float32), cuda)
    gy, gq = ops.macro_state_from_ru(gr, gu, um)
    ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()
    desc = ops.macro_desc(L, N, dt, dx, um)
    res = []
    try:
        for v, w in ((0, 1), (variant, waves)):
            assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, v) == 0
            assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, w) == 0
            tape = torch.zeros(ops.macro_tape_numel(desc, T), device=cuda)
            hist = torch.zeros(T, L, 3, N, device=cuda)
            out = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape, hist=hist)
            # the raw tape lists its exceptions in the order the forward solved them (that depends on the split and on the
            # kernel); what must not depend on anything is what the reverse sweep makes of it: the blocks
            res.append((out, ops.macro_tape_expand(desc, T, tape), hist))
    finally:
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, 0)
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, 0)
    if variant == 0:
        for a, b in zip(res[0][0], res[1][0]):
            assert torch.equal(a, b)
        assert torch.equal(res[0][2], res[1][2])
        assert torch.equal(res[0][1], res[1][1])
    else:
        for a, b in zip(res[0][0], res[1][0]):
            assert rel_max(b.cpu().numpy(), a.cpu().numpy()) <= 1e-6
        assert rel_max(res[1][2].cpu().numpy(), res[0][2].cpu().numpy()) <= 1e-6
        assert rel_max(res[1][1].cpu().numpy(), res[0][1].cpu().numpy()) <= 1e-6
        # the first step's tape sees identical inputs: bitwise
        assert torch.equal(res[0][1][0], res[1][1][0])


def test_macro_two_phase_glue_matches_ieee_glue(cuda):
    """One step of the two kernels on many random cells: the (u, u_eq) the two-phase kernel derives from its double-precision
    square roots equal the IEEE float32 glue of the one-phase kernel up to 1 ulp, on all but a vanishing share of cells."""
    import torch
    from dhts import _lib, ops
    from util import ulp_diff
    rng = np.random.default_rng(8)
    L, N, dt, dx, um = 64, 1000, 0.01, 5.0, 30.0
    r0 = rng.uniform(0.0, 1.2, (L, N)).astype(np.float32)
    r0[:, ::97] = 0.0
    r0[:, 5::101] = rng.uniform(0.0, 2e-5, r0[:, 5::101].shape).astype(np.float32)
    r = T_(r0, cuda)
    u = T_(rng.uniform(0.0, um, (L, N)).astype(np.float32), cuda)
    y, q = ops.macro_state_from_ru(r, u, um)
    gr = T_(rng.uniform(0.0, 1.0, (L, 2)).astype(np.float32), cuda)
    gu = T_(rng.uniform(0.0, um, (L, 2)).astype(np.float32), cuda)
    gy, gq = ops.macro_state_from_ru(gr, gu, um)
    ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()
    desc = ops.macro_desc(L, N, dt, dx, um)
    outs = []
    try:
        for v in (0, 1):
            assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, v) == 0
            outs.append([t.cpu().numpy() for t in ops.macro_rollout_fwd(desc, 1, r, y, u, q, ghost)])
    finally:
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, 0)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])      # r, y: same solver code
    for k in (2, 3):
        d = ulp_diff(outs[0][k], outs[1][k])
        assert d.max() <= 1 and (d > 0).mean() <= 1e-4, (k, int(d.max()), float((d > 0).mean()))


@pytest.mark.parametrize("T", [0, 1, 2])
def test_macro_rollout_short_horizons(cuda, oracle, T):
    """T = 0 returns the input state; T = 1, 2 exercise the first / last iteration of the two-phase kernel's step loop."""
    import torch
    from dhts import ops
    rng = np.random.default_rng(3)
    L, N, dt, dx, um = 3, 130, 0.01, 5.0, 30.0
    r0 = rng.uniform(0.05, 0.95, (L, N)).astype(np.float32)
    u0 = rng.uniform(0.0, um, (L, N)).astype(np.float32)
    gr = rng.uniform(0.05, 0.95, (L, 2)).astype(np.float32)
    gu = rng.uniform(0.0, um, (L, 2)).astype(np.float32)
    r, u = T_(r0, cuda), T_(u0, cuda)
    y, q = ops.macro_state_from_ru(r, u, um)
    tgr, tgu = T_(gr, cuda), T_(gu, cuda)
    gy, gq = ops.macro_state_from_ru(tgr, tgu, um)
    ghost = torch.stack([tgr, gy, tgu, gq], dim=-1).contiguous()
    desc = ops.macro_desc(L, N, dt, dx, um)
    tape = torch.zeros(max(ops.macro_tape_numel(desc, T), 1), device=cuda)
    out = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape)
    if T == 0:
        for a, b in zip(out, (r, y, u, q)):
            assert torch.equal(a, b)
        return
    f = oracle.macro_rollout_fwd(r0, u0, gr, gu, T, dt, dx, um)
    assert rel_max(out[0].cpu().numpy(), f["rT"]) <= 1e-6 and rel_max(out[2].cpu().numpy(), f["uT"]) <= 1e-6


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("N,T", [(2, 5), (64, 9), (130, 31), (256, 16), (512, 23), (700, 14), (1024, 11), (1025, 4), (1026, 6),
                                 (1100, 7), (1500, 1), (1500, 2), (1500, 9), (2047, 5), (2048, 12), (2049, 3)])
def test_macro_reverse_sweeps_agree_on_one_tape(cuda, N, T, variant):
    """The rollout's ways from a tape to a gradient give the same bits: the pipelined one-cell-per-thread kernel (every
    block size up to 1024 cells, with and without per-step cotangents; tapes of the two-phase forward kernel and -- all
    interfaces exceptions, more of them than threads -- of the one-phase kernel), the pipelined two-cells-per-thread kernel
    (1026 .. 2048 cells without per-step cotangents), the general kernel (everything else),
    and the single-step operator's sweep over the blocks dhts_macro_tape_expand writes out, one step at a time.  (Against the oracle:
    test_macro_rollout_vs_oracle_sizes and the goldens.)"""
    import torch
    from dhts import _lib, ops
    rng = np.random.default_rng(100 + N)
    L, dt, dx, um = 5, 0.01, 5.0, 30.0
    r0 = rng.uniform(0.0, 1.0, (L, N)).astype(np.float32)
    r0[1, : min(N, 3)] = 0.0                       # vacuum at the left boundary
    u0 = rng.uniform(0.0, um, (L, N)).astype(np.float32)
    gr = rng.uniform(0.05, 0.95, (L, 2)).astype(np.float32)
    gu = rng.uniform(0.0, um, (L, 2)).astype(np.float32)
    r, u = T_(r0, cuda), T_(u0, cuda)
    y, q = ops.macro_state_from_ru(r, u, um)
    tgr, tgu = T_(gr, cuda), T_(gu, cuda)
    gy, gq = ops.macro_state_from_ru(tgr, tgu, um)
    ghost = torch.stack([tgr, gy, tgu, gq], dim=-1).contiguous()
    desc = ops.macro_desc(L, N, dt, dx, um)
    tape = torch.full((ops.macro_tape_numel(desc, T),), float("nan"), device=cuda)     # unwritten parts must never be read
    try:
        assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, variant) == 0
        rT, yT, uT, _ = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape)
    finally:
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, 0)
    g_r, g_y = 2 * rT, torch.zeros_like(rT)
    ops.macro_u_tap_bwd(rT, yT, 2 * uT, g_r, g_y, um)
    plan = ops.macro_rollout_plan(desc, T)
    assert plan["bwd_pipelined"] == (1 if N <= 1024 else (2 if 1026 <= N <= 2048 else 0))
    assert ops.macro_rollout_plan(desc, T, want_hist=True)["bwd_pipelined"] == (1 if N <= 1024 else 0)
    fast = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y)
    general = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y, g_hist=torch.zeros(T, L, 2, N, device=cuda))
    assert torch.equal(fast[0], general[0]) and torch.equal(fast[1], general[1]) and torch.equal(fast[2], general[2])
    dqs = ops.macro_tape_expand(desc, T, tape)                 # [T][L][3][Np][4]
    assert torch.isfinite(dqs).all()
    a, b = g_r.clone(), g_y.clone()
    gh = torch.zeros(L, 2, 2, dtype=torch.float64, device=cuda)
    for t in range(T - 1, -1, -1):
        a, b, g1 = ops.macro_step_bwd(desc, dqs[t].contiguous(), a, b)
        gh += g1
    assert torch.equal(fast[0], a) and torch.equal(fast[1], b)
    assert rel_max(fast[2].cpu().numpy(), gh.cpu().numpy()) <= 1e-6      # (the ghosts' sums differ in the order of their doubles)
    # per-step cotangents (a loss on the state history): the sweep adds g_hist[t] in front of step t
    g_hist = torch.randn(T, L, 2, N, device=cuda)
    with_hist = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y, g_hist=g_hist)
    a, b = g_r.clone(), g_y.clone()
    for t in range(T - 1, -1, -1):
        a, b, _ = ops.macro_step_bwd(desc, dqs[t].contiguous(), a + g_hist[t, :, 0], b + g_hist[t, :, 1])
    assert torch.equal(with_hist[0], a) and torch.equal(with_hist[1], b)


def test_macro_tape_matches_oracle_over_rollout(cuda, oracle):
    """The whole Jacobian tape of a rollout, entry by entry."""
    import torch
    from dhts import ops
    rng = np.random.default_rng(5)
    L, N, T, dt, dx, um = 3, 150, 12, 0.01, 5.0, 30.0
    r0 = rng.uniform(0.05, 0.95, (L, N)).astype(np.float32)
    u0 = rng.uniform(0.0, um, (L, N)).astype(np.float32)
    gr = rng.uniform(0.05, 0.95, (L, 2)).astype(np.float32)
    gu = rng.uniform(0.0, um, (L, 2)).astype(np.float32)
    f = oracle.macro_rollout_fwd(r0, u0, gr, gu, T, dt, dx, um)
    desc = ops.macro_desc(L, N, dt, dx, um)
    r, u = T_(r0, cuda), T_(u0, cuda)
    y, q = ops.macro_state_from_ru(r, u, um)
    tgr, tgu = T_(gr, cuda), T_(gu, cuda)
    gy, gq = ops.macro_state_from_ru(tgr, tgu, um)
    ghost = torch.stack([tgr, gy, tgu, gq], dim=-1).contiguous()
    tape = torch.zeros(ops.macro_tape_numel(desc, T), device=cuda)
    ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape)
    dqs = rollout_tape_to_dqs(desc, tape, T)
    # a 1-ulp difference in a cell's state changes that cell's later tape entries in the last bits, so the
    # comparison is norm-relative; the single-step test above compares a tape bit for bit
    assert rel_max(dqs, f["tape"]) <= 2e-6
    for t in range(T):
        assert rel_max(dqs[t], f["tape"][t]) <= 2e-6


def test_macro_cfl_fault(cuda):
    """dt * speed >= dx: the reference asserts (_macro_lane.py:141-146); the drop-in raises the same way."""
    import dhts
    L, N = 2, 70
    r0 = T_(np.full((L, N), 0.3, np.float32), cuda)
    u0 = T_(np.full((L, N), 20.0, np.float32), cuda)
    gr = T_(np.full((L, 2), 0.3, np.float32), cuda)
    gu = T_(np.full((L, 2), 20.0, np.float32), cuda)
    with pytest.raises(AssertionError, match="CFL"):
        dhts.macro_rollout(r0, u0, gr, gu, 3, 1.0, 5.0, 30.0)
    dhts.macro_rollout(r0, u0, gr, gu, 3, 0.01, 5.0, 30.0)


@pytest.mark.parametrize("N,group", [(128, 1), (128, 4), (256, 2), (384, 4), (512, 1), (512, 2), (512, 4), (1024, 1)])
def test_macro_pair_kernel_equals_lane_kernel(cuda, N, group):
    """The pair kernel (a thread owns two adjacent cells and their right interfaces; what full lanes of 128 W cells launch) with
    1, 2 or 4 traffic lanes per workgroup against the lane kernel (DHTS_OPT_MACRO_FWD_VARIANT = 2, one lane per workgroup): final
    state, the blocks the tape expands to and the reverse sweep's gradients bit for bit, with and without a tape.  The lanes hold
    vacuum stretches (the general form of phase 1 beside the dense one), a density of exactly float32(eps), an idle lane (only the
    standing queue entries), a lane that queues nearly every interface (more entries than a round of phase 2 has threads) and
    an empty lane; a CFL fault names the lane it is in."""
    import torch
    from dhts import _lib, ops
    rng = np.random.default_rng(300 + N + group)
    L, T, dt, dx, um = 256 * group, 40, 0.01, 5.0, 30.0
    r0 = rng.uniform(0.02, 1.0, (L, N)).astype(np.float32)
    r0[1, 10:20] = 0.0
    r0[2] = 0.4                              # an idle lane
    r0[3, 64] = np.float32(1e-5)             # exactly float32(eps)
    r0[3, 65] = 1e-6
    r0[5, ::2] = 0.0                         # the longest queue
    r0[6] = 0.0                              # an empty lane
    r0[7, 126:130 if N > 128 else 128] = 0.0 # vacuum across a wavefront's chunk boundary
    r0[L - 1, N - 3:] = 0.0
    u0 = rng.uniform(0.0, um, (L, N)).astype(np.float32)
    u0[2] = 12.0
    r, u = T_(r0, cuda), T_(u0, cuda)
    y, q = ops.macro_state_from_ru(r, u, um)
    gr = T_(rng.uniform(0.0, 1.0, (L, 2)).astype(np.float32), cuda)
    gu = T_(rng.uniform(0.0, um, (L, 2)).astype(np.float32), cuda)
    gy, gq = ops.macro_state_from_ru(gr, gu, um)
    ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()
    desc = ops.macro_desc(L, N, dt, dx, um)
    res = []
    try:
        for variant, grp in ((2, 1), (0, group)):
            assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, variant) == 0
            assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, grp) == 0
            plan = ops.macro_rollout_plan(desc, T)
            assert plan["fwd_kernel"] == (2 if variant == 0 else 0) and plan["fwd_lanes_per_group"] == grp, plan
            if variant == 0:
                # what does not take the pair kernel: a state history, lanes that are not 128 W cells, a forced wave count
                assert ops.macro_rollout_plan(desc, T, want_hist=True)["fwd_kernel"] == 0
                assert ops.macro_rollout_plan(ops.macro_desc(L, N - 1, dt, dx, um), T)["fwd_kernel"] == 0
                assert ops.macro_rollout_plan(ops.macro_desc(L, N + 64, dt, dx, um), T)["fwd_kernel"] == 0
                # lanes that do not divide by the group take a smaller one
                assert ops.macro_rollout_plan(ops.macro_desc(L + 1, N, dt, dx, um), T)["fwd_lanes_per_group"] == 1
            tape = torch.full((ops.macro_tape_numel(desc, T),), float("nan"), device=cuda)      # unwritten parts must never be read
            out = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape)
            plain = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost)
            for a, b in zip(out, plain):
                assert torch.equal(a, b)
            g_r, g_y = 2 * out[0], torch.zeros_like(out[0])
            ops.macro_u_tap_bwd(out[0], out[1], 2 * out[2], g_r, g_y, um)
            res.append((out, ops.macro_tape_expand(desc, T, tape), ops.macro_rollout_bwd(desc, T, tape, g_r, g_y)))
        for a, b in zip(res[0][0], res[1][0]):
            assert torch.equal(a, b)
        assert torch.equal(res[0][1], res[1][1])
        for a, b in zip(res[0][2], res[1][2]):
            assert torch.equal(a, b)
        assert bool(torch.isfinite(res[1][0][0]).all())
        # a CFL fault in the last lane of a group, on an interface between two wavefronts' chunks, is reported for that lane
        bad = L - 2 * group - 1
        u_bad, r_bad = u.clone(), r.clone()
        c0 = 127 if N > 128 else 7
        r_bad[bad, c0:c0 + 2] = 0.3
        u_bad[bad, c0:c0 + 2] = 600.0                           # dt * speed >= dx
        y_bad, q_bad = ops.macro_state_from_ru(r_bad, u_bad, um)
        err = ops.new_error_record(cuda)
        ops.macro_rollout_fwd(desc, T, r_bad, y_bad, u_bad, q_bad, ghost, err=err)
        rec = err.cpu().numpy()
        assert rec[0] == _lib.FAULT_CFL and 0 <= rec[1] < T and rec[2] == bad and 0 <= rec[3] <= N, rec
    finally:
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, 0)
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, 0)


def test_macro_long_lane_reverse_sweep_is_repeatable(cuda):
    """256 lanes x 2048 cells x 300 steps from random cells (the first steps queue more exceptions than the reverse sweep has
    threads): the two-cells-per-thread reverse sweep twice and the general kernel once give the same bits.  (A missing
    barrier behind the sweep's prologue showed up here as a difference in a few lanes per run.)"""
    import torch
    from dhts import ops
    L, N, T, dt, dx, um = 256, 2048, 300, 0.01, 5.0, 30.0
    gen = torch.Generator().manual_seed(5)
    r = (0.05 + 0.9 * torch.rand(L, N, generator=gen)).to(cuda)
    u = (um * torch.rand(L, N, generator=gen)).to(cuda)
    gr = (0.05 + 0.9 * torch.rand(L, 2, generator=gen)).to(cuda)
    gu = (um * torch.rand(L, 2, generator=gen)).to(cuda)
    y, q = ops.macro_state_from_ru(r, u, um)
    gy, gq = ops.macro_state_from_ru(gr, gu, um)
    ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()
    desc = ops.macro_desc(L, N, dt, dx, um)
    assert ops.macro_rollout_plan(desc, T)["bwd_pipelined"] == 2
    tape = torch.empty(ops.macro_tape_numel(desc, T), device=cuda)
    out = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape)
    g_r, g_y = 2 * out[0], torch.zeros_like(out[0])
    zeros = torch.zeros(T, L, 2, N, device=cuda)
    general = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y, g_hist=zeros)
    for _ in range(4):
        fast = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y)
        for a, b in zip(fast, general):
            assert torch.equal(a, b)


def test_macro_full_size_properties(cuda):
    """BASELINE config 2 (1024 lanes x 512 cells x 1000 steps) through size-independent properties:
    bitwise repeatability, replica consistency (identical lanes -> identical results), lane independence
    (a lane's result does not depend on its neighbours in the batch), and exact homogeneity of the adjoint."""
    import torch
    from dhts import ops
    L, N, T, dt, dx, um = 1024, 512, 1000, 0.01, 5.0, 30.0
    gen = torch.Generator(device="cpu").manual_seed(2026)
    r0 = (0.05 + 0.9 * torch.rand(L, N, generator=gen)).to(cuda)
    u0 = (um * torch.rand(L, N, generator=gen)).to(cuda)
    gr = (0.05 + 0.9 * torch.rand(L, 2, generator=gen)).to(cuda)
    gu = (um * torch.rand(L, 2, generator=gen)).to(cuda)
    # lanes 1 and 2 are copies of lane 0; lane 3 is lane 700's copy
    for dst, src in ((1, 0), (2, 0), (3, 700)):
        r0[dst], u0[dst], gr[dst], gu[dst] = r0[src], u0[src], gr[src], gu[src]
    desc = ops.macro_desc(L, N, dt, dx, um)
    y0, q0 = ops.macro_state_from_ru(r0, u0, um)
    gy, gq = ops.macro_state_from_ru(gr, gu, um)
    ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()
    tape = torch.empty(ops.macro_tape_numel(desc, T), device=cuda)
    err = ops.new_error_record(cuda)
    out1 = ops.macro_rollout_fwd(desc, T, r0, y0, u0, q0, ghost, tape=tape, err=err)
    assert ops.raise_on_fault(err) == 0
    rT, yT, uT, qT = out1
    assert torch.isfinite(rT).all() and torch.isfinite(uT).all()
    g_r, g_y = 2 * rT, torch.zeros_like(rT)
    ops.macro_u_tap_bwd(rT, yT, 2 * uT, g_r, g_y, um)
    b1 = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y)
    # replica consistency
    for dst, src in ((1, 0), (2, 0), (3, 700)):
        assert torch.equal(rT[dst], rT[src]) and torch.equal(uT[dst], uT[src])
        assert torch.equal(b1[0][dst], b1[0][src]) and torch.equal(b1[1][dst], b1[1][src])
    # exact homogeneity: J^T (2 g) == 2 J^T g bit for bit (power-of-two scaling is exact in float32)
    b2 = ops.macro_rollout_bwd(desc, T, tape, 2 * g_r, 2 * g_y)
    assert torch.equal(b2[0], 2 * b1[0]) and torch.equal(b2[1], 2 * b1[1])
    # bitwise repeatability of forward + tape
    tape2 = torch.empty_like(tape)
    out2 = ops.macro_rollout_fwd(desc, T, r0, y0, u0, q0, ghost, tape=tape2)
    assert all(torch.equal(a, b) for a, b in zip(out1, out2))
    # (two raw tapes may list their exceptions in different orders: what the reverse sweep makes of them must be the same)
    b3 = ops.macro_rollout_bwd(desc, T, tape2, g_r, g_y)
    assert torch.equal(b3[0], b1[0]) and torch.equal(b3[1], b1[1])
    del tape2
    # lane independence: a 16-lane sub-batch reproduces the same lanes of the full batch
    sel = slice(690, 706)
    d16 = ops.macro_desc(16, N, dt, dx, um)
    tape16 = torch.empty(ops.macro_tape_numel(d16, T), device=cuda)
    o16 = ops.macro_rollout_fwd(d16, T, r0[sel].contiguous(), y0[sel].contiguous(), u0[sel].contiguous(), q0[sel].contiguous(),
                                ghost[sel].contiguous(), tape=tape16)
    assert torch.equal(o16[0], rT[sel]) and torch.equal(o16[2], uT[sel])


BENCH_LANES = (0, 1, 2, 3, 500, 511, 1022, 1023)      # lanes of bench.py's rank-0 config-2 / config-3 tensors that meet the oracle


def test_macro_bench_instantiation_vs_reference_and_oracle(cuda, oracle, golden_dir):
    """The kernel instantiations bench.py times on BASELINE config 2 -- macro_rollout_fwd3_kernel<1, true> (the pair kernel: one
    traffic lane per workgroup at four wavefronts per lane, two adjacent cells per thread, no history) and
    macro_rollout_bwd_fast_kernel<512, false> -- in
    the bench's own launch: 1024 lanes x 512 cells x 1000 steps, the tensors bench.py builds for rank 0.  Lane 4 of the batch is
    replaced by the reference's own 512 x 1000 run (golden c2slice: state <= 1e-5, gradients <= 1e-4); eight more lanes are
    compared with the oracle over all 1000 steps."""
    import torch
    import bench
    import dhts
    from dhts import ops
    g = load(golden_dir, "macro_rollout_c2slice.npz")
    m = meta_of(g)
    N, T, dt, dx, um = m["N"], m["T"], m["dt"], m["dx"], m["u_max"]
    assert (N, T, dt, dx, um) == (512, 1000, 0.01, 5.0, 30.0)
    # the whole batch bench.py builds for rank 0 (1024 lanes), lane 4 replaced by the reference's lane
    r0, u0, gr, gu = (t.numpy().astype(np.float32).copy() for t in bench.MacroWorkload.inputs(0, 1024, 512, um))
    k = 4
    r0[k], u0[k], gr[k], gu[k] = g["r0"], g["u0"], g["ghost_r"], g["ghost_u"]
    pick = sorted(set(BENCH_LANES) | {k})
    L = r0.shape[0]
    # the launch this test makes is the one the bench makes
    plan = ops.macro_rollout_plan(ops.macro_desc(L, N, dt, dx, um), T, want_hist=False)
    assert plan == dict(fwd_kernel=2, fwd_waves=4, fwd_passes=2, fwd_full_lane=1, bwd_pipelined=1, bwd_block=512, hist=0,
                        fwd_lanes_per_group=1)
    tr0, tu0 = T_(r0, cuda, grad=True), T_(u0, cuda, grad=True)
    tgr, tgu = T_(gr, cuda, grad=True), T_(gu, cuda, grad=True)
    rT, yT, uT, _ = dhts.macro_rollout(tr0, tu0, tgr, tgu, T, dt, dx, um)          # no history: the bench's instantiation
    ((rT ** 2).sum() + (uT ** 2).sum()).backward()
    rT, yT, uT = (t.detach().cpu().numpy() for t in (rT, yT, uT))
    g_r0, g_u0, g_gr, g_gu = (t.grad.cpu().numpy() for t in (tr0, tu0, tgr, tgu))
    # the reference's lane
    assert rel_max(rT[k], g["rT"]) <= TOL_STATE and rel_max(yT[k], g["yT"]) <= TOL_STATE and rel_max(uT[k], g["uT"]) <= TOL_STATE
    e_elem = max(rel_elem(rT[k], g["rT"]), rel_elem(uT[k], g["uT"]))
    print("G4 c2slice (bench instantiation): element-wise state error %.2e" % e_elem)
    assert e_elem <= TOL_STATE
    assert grad_report("G4 c2slice (bench instantiation) d loss / d r0", g_r0[k], g["g_r0"]) <= TOL_GRAD
    assert grad_report("G4 c2slice (bench instantiation) d loss / d u0", g_u0[k], g["g_u0"]) <= TOL_GRAD
    assert rel_max(g_gr[k], g["g_ghost_r"]) <= TOL_GRAD and rel_max(g_gu[k], g["g_ghost_u"]) <= TOL_GRAD
    # config 2's own lanes against the oracle, all 1000 steps
    f = oracle.macro_rollout_fwd(r0[pick], u0[pick], gr[pick], gu[pick], T, dt, dx, um)
    assert f["rc"] == 0
    b = oracle.macro_rollout_bwd(f, g_rT=2 * f["rT"], g_uT=2 * f["uT"])
    worst_s = worst_g = 0.0
    for o, j in enumerate(pick):
        worst_s = max(worst_s, rel_elem(rT[j], f["rT"][o]), rel_elem(uT[j], f["uT"][o]), rel_max(yT[j], f["yT"][o]))
        worst_g = max(worst_g, rel_max(g_r0[j], b["g_r0"][o]), rel_max(g_u0[j], b["g_u0"][o]))
        assert rel_max(g_gr[j], b["g_ghost_r"][o]) <= TOL_GRAD and rel_max(g_gu[j], b["g_ghost_u"][o]) <= TOL_GRAD
    print("config 2 lanes %s (bench instantiation) vs oracle over %d steps: state %.2e (element-wise), gradient %.2e" % (
        pick, T, worst_s, worst_g))
    assert worst_s <= TOL_STATE and worst_g <= TOL_GRAD


# =================================================================================================================
# micro
# =================================================================================================================
def micro_inputs(g, cuda):
    import torch
    par = g["params"]                                        # [V][6]
    params = T_(np.ascontiguousarray(par.T[:, None, :]), cuda, dtype=torch.float64)    # [6][1][V]
    return params


@pytest.mark.parametrize("name", ["inv10", "rand24", "dense16", "long", "x2", "x9", "x15", "x23", "x37", "x44"])   # x<k>: random shapes (gen_goldens --only G6x,Gx_pick)
def test_micro_rollout_vs_golden(cuda, golden_dir, name):
    import torch
    import dhts
    g = load(golden_dir, "micro_rollout_%s.npz" % name)
    m = meta_of(g)
    p0, v0 = T_(g["p0"][None], cuda, grad=True), T_(g["v0"][None], cuda, grad=True)
    head = T_(np.array([m["head"]], dtype=np.float64), cuda)
    out = dhts.micro_rollout(p0, v0, micro_inputs(g, cuda), head, m["T"], m["dt"], want_hist=True)
    pT, vT, hist = out
    if m["tap"] == "every_sum":
        loss = hist.sum()
    else:
        loss = 1e-4 * (pT ** 2).sum() + (vT ** 2).sum()
    loss.backward()
    assert rel_max(pT.detach().cpu().numpy()[0], g["pT"]) <= 1e-6
    assert rel_max(vT.detach().cpu().numpy()[0], g["vT"]) <= 1e-6
    assert rel_elem(pT.detach().cpu().numpy()[0], g["pT"]) <= 1e-6 and rel_elem(vT.detach().cpu().numpy()[0], g["vT"]) <= 1e-6
    h = hist.detach().cpu().numpy()
    for t in range(len(g["steps_p"])):
        assert ulp_diff(h[t, 0, 0], g["steps_p"][t]).max() <= 1
        assert ulp_diff(h[t, 0, 1], g["steps_v"][t]).max() <= 1
    assert grad_report("G6 %s (kernels) d loss / d p0" % name, p0.grad.cpu().numpy()[0], g["g_p0"]) <= 1e-5
    assert grad_report("G6 %s (kernels) d loss / d v0" % name, v0.grad.cpu().numpy()[0], g["g_v0"]) <= 1e-5


@pytest.mark.parametrize("V", [1, 2, 63, 64, 65, 128, 200, 256, 300, 600, 1024])
def test_micro_rollout_vs_oracle_sizes(cuda, oracle, V):
    import torch
    import dhts
    rng = np.random.default_rng(200 + V)
    L, T, dt = 4, 30, 0.01
    p0 = (np.arange(V)[None, :] * 20.0 + rng.uniform(0, 10, (L, V))).astype(np.float32)
    v0 = rng.uniform(9, 21, (L, V)).astype(np.float32)
    par = np.empty((L, V, 6))
    par[..., 0] = 30.0 * rng.uniform(0.8, 1.5, (L, V))
    par[..., 1] = 30.0 * rng.uniform(0.6, 1.5, (L, V))
    par[..., 2] = 30.0 * rng.uniform(0.8, 1.2, (L, V))
    par[..., 3] = rng.uniform(0.5, 5.0, (L, V))
    par[..., 4] = rng.uniform(0.1, 1.5, (L, V))
    par[..., 5] = 5.0
    f = oracle.micro_rollout_fwd(p0, v0, par, T, dt, 37.5, 1.25)
    w_p = rng.standard_normal((L, V)).astype(np.float32)
    w_v = rng.standard_normal((L, V)).astype(np.float32)
    b = oracle.micro_rollout_bwd(f, g_pT=w_p, g_vT=w_v)
    tp0, tv0 = T_(p0, cuda, grad=True), T_(v0, cuda, grad=True)
    head = T_(np.tile([37.5, 1.25], (L, 1)), cuda, dtype=torch.float64, grad=True)
    params = T_(par.transpose(2, 0, 1), cuda, dtype=torch.float64)
    pT, vT = dhts.micro_rollout(tp0, tv0, params, head, T, dt)
    ((pT * T_(w_p, cuda)).sum() + (vT * T_(w_v, cuda)).sum()).backward()
    assert rel_max(pT.detach().cpu().numpy(), f["pT"]) <= 1e-6
    assert rel_max(vT.detach().cpu().numpy(), f["vT"]) <= 1e-6
    assert rel_max(tp0.grad.cpu().numpy(), b["g_p0"]) <= 1e-5
    assert rel_max(tv0.grad.cpu().numpy(), b["g_v0"]) <= 1e-5
    assert rel_max(head.grad.cpu().numpy(), b["g_head"]) <= 1e-4


@pytest.fixture
def micro_fwd_waves(request):
    """DHTS_OPT_MICRO_FWD_WAVES for one test: 1 / 2 / 4 wavefronts per lane in the micro forward kernel (0 = heuristic)."""
    from dhts import _lib
    assert _lib.lib().dhts_set_option(_lib.OPT_MICRO_FWD_WAVES, request.param) == 0
    yield request.param
    _lib.lib().dhts_set_option(_lib.OPT_MICRO_FWD_WAVES, 0)


@pytest.mark.parametrize("T", [20, 21])
@pytest.mark.parametrize("micro_fwd_waves", [0, 1, 2, 4], indirect=True)
def test_micro_ragged_and_empty_lanes(cuda, oracle, micro_fwd_waves, T):
    """Per-lane vehicle counts: empty lane, single vehicle, partly filled, full -- for every wavefronts-per-lane variant of
    the forward kernel, each with an even and an odd T (the state ends in either ping-pong buffer; slots beyond the count
    must pass through from the input in both)."""
    import torch
    from dhts import ops
    rng = np.random.default_rng(9)
    V, dt = 200, 0.01
    counts = [0, 1, 37, 64, 65, 128, 129, 200]
    L = len(counts)
    p0 = (np.arange(V)[None, :] * 20.0 + rng.uniform(0, 10, (L, V))).astype(np.float32)
    v0 = rng.uniform(9, 21, (L, V)).astype(np.float32)
    par = np.tile(np.array([30.0, 24.0, 27.0, 0.5, 0.1, 5.0]), (L, V, 1))
    desc = ops.micro_desc(L, V, dt)
    tape = torch.zeros(ops.micro_tape_numel(desc, T), device=cuda)
    head = T_(np.tile([1000.0, 0.0], (L, 1)), cuda, dtype=torch.float64)
    params = T_(par.transpose(2, 0, 1), cuda, dtype=torch.float64)
    count = T_(np.array(counts, np.int32), cuda)
    pT, vT = ops.micro_rollout_fwd(desc, T, T_(p0, cuda), T_(v0, cuda), params, head, count=count, tape=tape)
    w = rng.standard_normal((2, L, V)).astype(np.float32)
    g_p0, g_v0, g_head = ops.micro_rollout_bwd(desc, T, tape, T_(w[0], cuda), T_(w[1], cuda), count=count)
    for l, n in enumerate(counts):
        if n == 0:
            assert torch.equal(pT[l].cpu(), torch.tensor(p0[l])) and torch.equal(vT[l].cpu(), torch.tensor(v0[l]))
            assert float(g_p0[l].abs().max()) == 0.0
            continue
        f = oracle.micro_rollout_fwd(p0[l:l + 1, :n], v0[l:l + 1, :n], par[l:l + 1, :n], T, dt)
        b = oracle.micro_rollout_bwd(f, g_pT=w[0, l:l + 1, :n], g_vT=w[1, l:l + 1, :n])
        assert rel_max(pT[l, :n].cpu().numpy(), f["pT"][0]) <= 1e-6
        assert rel_max(vT[l, :n].cpu().numpy(), f["vT"][0]) <= 1e-6
        assert torch.equal(pT[l, n:].cpu(), torch.tensor(p0[l, n:]))      # untouched slots pass through
        assert torch.equal(vT[l, n:].cpu(), torch.tensor(v0[l, n:]))
        assert rel_max(g_p0[l, :n].cpu().numpy(), b["g_p0"][0]) <= 1e-5
        assert rel_max(g_v0[l, :n].cpu().numpy(), b["g_v0"][0]) <= 1e-5
        assert float(g_p0[l, n:].abs().max()) == 0.0 if n < V else True


def test_micro_collision_is_recorded_and_tolerated(cuda, oracle):
    """gap < 0: the reference prints, zeroes the deltas and carries on (_micro_lane.py:151-162)."""
    import torch
    from dhts import ops
    par = np.tile(np.array([30.0, 24.0, 27.0, 0.5, 0.1, 5.0]), (1, 2, 1))
    p0 = np.array([[0.0, 3.0]], np.float32)
    v0 = np.array([[10.0, 1.0]], np.float32)
    desc = ops.micro_desc(1, 2, 0.01)
    err = ops.new_error_record(cuda)
    tape = torch.zeros(ops.micro_step_tape_numel(desc), device=cuda)
    head = T_(np.array([[1000.0, 0.0]]), cuda, dtype=torch.float64)
    pT, vT = ops.micro_step_fwd(desc, T_(p0, cuda), T_(v0, cuda), T_(par.transpose(2, 0, 1), cuda, dtype=torch.float64),
                                head, tape=tape, err=err)
    code, step, lane, index = err.tolist()
    assert (code, step, lane, index) == (2, 0, 0, 0)
    o = oracle.micro_step(p0[0], v0[0], par[0], 1000.0, 0.0, 0.01)
    assert rel_max(pT.cpu().numpy()[0], o["np"]) <= 1e-6 and rel_max(vT.cpu().numpy()[0], o["nv"]) <= 1e-6
    dqs = tape_to_dqs(tape, 1, 1, 2, planes=2)[0, 0]
    assert rel_max(dqs, o["dqs"]) <= 1e-5


def test_micro_full_size_properties(cuda):
    """BASELINE config 3 (4096 lanes x 256 vehicles = 2^20, 1000 steps): repeatability, replica consistency,
    exact homogeneity of the adjoint, and ordering (no vehicle overtakes its leader)."""
    import torch
    from dhts import ops
    L, V, T, dt = 4096, 256, 1000, 0.01
    gen = torch.Generator(device="cpu").manual_seed(7)
    p0 = (torch.arange(V)[None, :] * 20.0 + 10.0 * torch.rand(L, V, generator=gen)).to(cuda)
    v0 = (9.0 + 12.0 * torch.rand(L, V, generator=gen)).to(cuda)
    p0[1], v0[1] = p0[0], v0[0]
    params = torch.tensor([30.0, 24.0, 27.0, 0.5, 0.1, 5.0], dtype=torch.float64, device=cuda)[:, None, None].expand(6, L, V).contiguous()
    head = torch.tensor([[1000.0, 0.0]], dtype=torch.float64, device=cuda).expand(L, 2).contiguous()
    desc = ops.micro_desc(L, V, dt)
    tape = torch.empty(ops.micro_tape_numel(desc, T), device=cuda)
    err = ops.new_error_record(cuda)
    pT, vT = ops.micro_rollout_fwd(desc, T, p0, v0, params, head, tape=tape, err=err)
    assert err.tolist()[0] == 0
    assert torch.isfinite(pT).all() and torch.isfinite(vT).all() and (vT >= 0).all()
    assert (pT[:, 1:] - pT[:, :-1] > 5.0).all()           # gaps stay larger than a vehicle length
    assert torch.equal(pT[1], pT[0]) and torch.equal(vT[1], vT[0])
    g1 = ops.micro_rollout_bwd(desc, T, tape, 2e-4 * pT, 2 * vT)
    g2 = ops.micro_rollout_bwd(desc, T, tape, 4e-4 * pT, 4 * vT)
    assert torch.equal(g2[0], 2 * g1[0]) and torch.equal(g2[1], 2 * g1[1])
    assert torch.equal(g1[0][1], g1[0][0])
    tape2 = torch.empty_like(tape)
    pT2, vT2 = ops.micro_rollout_fwd(desc, T, p0, v0, params, head, tape=tape2)
    assert torch.equal(pT, pT2) and torch.equal(vT, vT2) and torch.equal(tape, tape2)


def test_micro_bench_instantiation_vs_reference_and_oracle(cuda, oracle, golden_dir):
    """The instantiations bench.py times on BASELINE config 3 -- micro_rollout_fwd_kernel<2, 2, true, true> (two wavefronts x
    two passes, full lane) and the one-vehicle-per-thread reverse path -- at the bench's shape: 256 vehicles x 1000 steps.
    Lane 4 is the reference's own 256 x 1000 run (golden c3slice; the oracle reproduces it bit for bit); the other eight are
    lanes of bench.py's rank-0 tensors against the oracle over all 1000 steps."""
    import torch
    import bench
    import dhts
    from dhts import ops
    g = load(golden_dir, "micro_rollout_c3slice.npz")
    m = meta_of(g)
    V, T, dt = m["V"], m["T"], m["dt"]
    assert (V, T, dt) == (256, 1000, 0.01) and np.allclose(g["params"], np.array(bench.MicroWorkload.PARAMS)[None, :])
    p0b, v0b = (t.numpy() for t in bench.MicroWorkload.inputs(0, 4096, 256))
    pick = list(BENCH_LANES)
    p0 = np.concatenate([p0b[pick[:4]], g["p0"][None], p0b[pick[4:]]]).astype(np.float32)
    v0 = np.concatenate([v0b[pick[:4]], g["v0"][None], v0b[pick[4:]]]).astype(np.float32)
    L = p0.shape[0]
    plan = ops.micro_rollout_plan(ops.micro_desc(L, V, dt), T, has_count=False)
    assert plan == ops.micro_rollout_plan(ops.micro_desc(4096, V, dt), T, has_count=False)
    assert plan == dict(fwd_waves=2, fwd_passes=2, fwd_full_lane=1, bwd_one_vehicle_per_thread=1, bwd_block=256)
    par = np.tile(np.array(bench.MicroWorkload.PARAMS), (L, V, 1))
    params = T_(np.ascontiguousarray(par.transpose(2, 0, 1)), cuda, dtype=torch.float64)
    head = torch.tensor([[1000.0, 0.0]] * L, dtype=torch.float64, device=cuda)
    tp0, tv0 = T_(p0, cuda, grad=True), T_(v0, cuda, grad=True)
    pT, vT = dhts.micro_rollout(tp0, tv0, params, head, T, dt)
    (1e-4 * (pT ** 2).sum() + (vT ** 2).sum()).backward()
    pT, vT = pT.detach().cpu().numpy(), vT.detach().cpu().numpy()
    g_p0, g_v0 = tp0.grad.cpu().numpy(), tv0.grad.cpu().numpy()
    k = 4
    assert rel_elem(pT[k], g["pT"]) <= TOL_STATE and rel_elem(vT[k], g["vT"]) <= TOL_STATE
    assert grad_report("G6 c3slice (bench instantiation) d loss / d p0", g_p0[k], g["g_p0"]) <= TOL_GRAD
    assert grad_report("G6 c3slice (bench instantiation) d loss / d v0", g_v0[k], g["g_v0"]) <= TOL_GRAD
    f = oracle.micro_rollout_fwd(p0, v0, par, T, dt)
    assert f["rc"] == 0
    b = oracle.micro_rollout_bwd(f, g_pT=np.float32(2e-4) * f["pT"], g_vT=2 * f["vT"])
    e_s = max(rel_elem(pT, f["pT"]), rel_elem(vT, f["vT"]))
    e_g = max(rel_max(g_p0, b["g_p0"]), rel_max(g_v0, b["g_v0"]))
    print("config 3 lanes %s (bench instantiation) vs oracle over %d steps: state %.2e (element-wise), gradient %.2e" % (pick, T, e_s, e_g))
    assert e_s <= TOL_STATE and e_g <= TOL_GRAD


def test_micro_zero_step_backward(cuda):
    """T = 0: the reverse sweep has no tape (a NULL pointer from an empty tensor) and must hand the cotangents through."""
    import torch
    from dhts import ops
    L, V = 3, 200
    desc = ops.micro_desc(L, V, 0.01)
    g_p, g_v = torch.randn(L, V, device=cuda), torch.randn(L, V, device=cuda)
    tape = torch.empty(0, device=cuda)
    err = ops.new_error_record(cuda)
    o_p, o_v, g_head = ops.micro_rollout_bwd(desc, 0, tape, g_p, g_v, err=err)
    torch.cuda.synchronize()
    assert torch.equal(o_p, g_p) and torch.equal(o_v, g_v) and float(g_head.abs().max()) == 0.0 and err.tolist()[0] == 0
    assert ops.micro_rollout_plan(desc, 0)["bwd_one_vehicle_per_thread"] == 0


def test_micro_reverse_sweep_reports_non_finite_gradient(cuda):
    """A non-finite cotangent in the micro reverse sweep is kept in the result (the reference's dMicroForwardLayer.backward
    has no assert, dmicro_lane.py:271-298) and recorded as DHTS_FAULT_NAN with its step / lane / vehicle."""
    import torch
    from dhts import _lib, ops
    L, V, T = 4, 128, 6
    desc = ops.micro_desc(L, V, 0.01)
    tape = torch.zeros(ops.micro_tape_numel(desc, T), device=cuda)
    t3 = tape.view(T, L, 128, 3)
    t3[3, 2, 17, 1] = float("inf")                  # dEgo[1][1] of vehicle 17, lane 2, step 3
    g = torch.ones(L, V, device=cuda)
    err = ops.new_error_record(cuda)
    o_p, o_v, _ = ops.micro_rollout_bwd(desc, T, tape, g, g.clone(), err=err)
    code, step, lane, index = err.tolist()
    assert (code, step, lane, index) == (_lib.FAULT_NAN, 3, 2, 17)
    assert not bool(torch.isfinite(o_v[2, 17]))


def test_micro_bwd_fault_reads_the_autograd_sweeps_record(cuda):
    """dhts.micro_rollout's reverse sweep does not read its fault record back (no host synchronisation per backward); ops.micro_bwd_fault()
    does, on demand: None after a finite sweep, (step, lane, vehicle) + a RuntimeWarning after one that met a non-finite cotangent, and the
    record is cleared between two backward passes over the same graph (retain_graph)."""
    import warnings
    import torch
    from dhts import ops
    L, V, T = 2, 64, 5
    p0 = (torch.arange(V, device=cuda, dtype=torch.float32) * 12.0).repeat(L, 1).requires_grad_(True)
    v0 = torch.full((L, V), 8.0, device=cuda, requires_grad=True)
    base = torch.tensor([2.0, 1.6, 30.0, 2.0, 1.5, 5.0], dtype=torch.float64, device=cuda)
    params = base[:, None, None].expand(6, L, V).contiguous()
    head = torch.tensor([[1000.0, 0.0]] * L, dtype=torch.float64, device=cuda)
    pT, vT = ops.micro_rollout(p0, v0, params, head, T, 0.05)
    loss = pT.sum() + vT.sum()
    loss.backward(retain_graph=True)
    assert ops.micro_bwd_fault() is None
    g_bad = torch.ones(L, V, device=cuda)
    g_bad[1, 7] = float("nan")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.autograd.backward([pT, vT], [torch.ones(L, V, device=cuda), g_bad], retain_graph=True)
        where = ops.micro_bwd_fault()
    assert where is not None and where[1] == 1 and where[0] == T - 1 and any(issubclass(x.category, RuntimeWarning) for x in w)
    loss.backward()                                   # a finite sweep over the same graph: the record starts clean
    assert ops.micro_bwd_fault(warn=False) is None


@pytest.mark.parametrize("V", [1, 5, 64, 97])
def test_micro_step_tensor_ladder_vs_oracle(cuda, oracle, V):
    """dhts_micro_step_fwd_tensor (the plain MicroLane's float32 tensor arithmetic: itscp `micro` mode) against the oracle's restatement
    of the same ladder: next state bit for bit, Jacobian blocks to float32 rounding; and it is NOT the analytic operator's ladder."""
    import torch
    from dhts import ops
    rng = np.random.default_rng(100 + V)
    L = 3
    p = np.sort(rng.uniform(0.0, 12.0 * V, (L, V)).astype(np.float32), axis=1)
    p += np.arange(V, dtype=np.float32)[None, :] * 6.0                       # gaps of at least a vehicle length
    v = rng.uniform(0.0, 25.0, (L, V)).astype(np.float32)
    prm = np.empty((L, V, 6)); prm[:] = [2.0, 1.6, 30.0, 2.0, 1.5, 5.0]
    prm[..., 2] += rng.uniform(-3, 3, (L, V))
    head = np.stack([rng.uniform(5.0, 60.0, L), rng.uniform(-3.0, 3.0, L)], axis=1)
    head = head.astype(np.float32).astype(np.float64)                          # (the head gap is a float32 tensor's value there)
    dt = 1.0 / 30.0
    desc = ops.micro_desc(L, V, dt)
    tape = torch.empty(ops.micro_step_tape_numel(desc), dtype=torch.float32, device=cuda)
    params_d = torch.tensor(np.ascontiguousarray(prm.transpose(2, 0, 1)), dtype=torch.float64, device=cuda)
    args = (desc, torch.tensor(p, device=cuda), torch.tensor(v, device=cuda), params_d, torch.tensor(head, device=cuda))
    np_t, nv_t = ops.micro_step_fwd(*args, tape=tape, tensor_ladder=True)
    np_a, nv_a = ops.micro_step_fwd(*args, tensor_ladder=False)
    Vp = (V + 63) // 64 * 64
    tp = tape.view(L, 2, Vp, 4).cpu().numpy()
    differs = 0
    for l in range(L):
        o = oracle.micro_step_f32(p[l], v[l], prm[l], head[l, 0], head[l, 1], dt)
        assert o["rc"] == 0
        assert np.array_equal(np_t[l].cpu().numpy(), o["np"]) and np.array_equal(nv_t[l].cpu().numpy(), o["nv"]), l
        dq = o["dqs"].reshape(V, 2, 4)
        assert rel_max(tp[l, 0, :V], dq[:, 0]) <= 1e-6 and rel_max(tp[l, 1, :V], dq[:, 1]) <= 1e-6
        differs += int((nv_a[l].cpu().numpy() != o["nv"]).sum())
    assert V == 1 or differs > 0


def test_micro_step_tensor_ladder_at_the_clamps(cuda, oracle):
    """The float32 tensor ladder where the forward clamps (found by tools/probes/fuzz_env.py, pinned by tests/golden/itscp_micro_jam_*):
    a collision (both deltas become the ints 0, 0), a gap of exactly 0 and one below POSITION_DELTA_EPS (max() picks the Python float),
    a leader that has fallen behind (abs() flips the sign).  Autograd differentiates those operations: the blocks are finite, the
    gap's entries are 0 where the gap is a constant -- device and restatement agree, and dIDM's un-clamped formulas do not apply."""
    import torch
    from dhts import ops
    L, V, dt = 1, 8, 1.0 / 30.0
    # vehicles tail -> head; lengths 5: gap_i = |p[i+1] - p[i]| - 5
    p = np.array([[0.0, 4.0, 9.0, 14.000004, 30.0, 22.0, 40.0, 60.0]], dtype=np.float32)     # gaps: -1 (collided), 0, 4e-6, 11, 3 (leader behind), 13, 15
    v = np.array([[8.0, 6.0, 5.0, 7.0, 9.0, 4.0, 10.0, 12.0]], dtype=np.float32)
    prm = np.empty((L, V, 6)); prm[:] = [2.0, 1.6, 30.0, 2.0, 1.5, 5.0]
    head = np.array([[3e-6, 1.0]], dtype=np.float32).astype(np.float64)                        # the head's gap below the epsilon too
    desc = ops.micro_desc(L, V, dt)
    tape = torch.empty(ops.micro_step_tape_numel(desc), dtype=torch.float32, device=cuda)
    params_d = torch.tensor(np.ascontiguousarray(prm.transpose(2, 0, 1)), dtype=torch.float64, device=cuda)
    np_t, nv_t = ops.micro_step_fwd(desc, torch.tensor(p, device=cuda), torch.tensor(v, device=cuda), params_d, torch.tensor(head, device=cuda),
                                    tape=tape, tensor_ladder=True)
    tp = tape.view(L, 2, 64, 4).cpu().numpy()[0, :, :V]
    o = oracle.micro_step_f32(p[0], v[0], prm[0], head[0, 0], head[0, 1], dt)
    dq = o["dqs"].reshape(V, 2, 4)
    assert np.array_equal(np_t[0].cpu().numpy(), o["np"]) and np.array_equal(nv_t[0].cpu().numpy(), o["nv"])
    assert np.isfinite(tp).all() and np.isfinite(dq).all()
    assert rel_max(tp[0], dq[:, 0]) <= 1e-6 and rel_max(tp[1], dq[:, 1]) <= 1e-6
    for i in (0, 1, 2, 7):          # collided, zero gap, below the epsilon, the head below the epsilon: the gap is a constant
        assert tp[0, i, 2] == 0.0 and tp[1, i, 2] == 0.0, i
    assert tp[1, 0, 3] == 0.0       # collided: the leader's speed is out of the step too
    assert tp[1, 3, 2] > 0.0 and tp[1, 4, 2] < 0.0 and tp[0, 4, 2] == -tp[1, 4, 2]       # a leader ahead / behind: the sign of abs()


@pytest.mark.gpu
def test_macro_step_reads_a_source_ghost_in_double(cuda):
    """A boundary cell of plain Python floats -- an itscp source lane's inflow (r, u_eq(r)), _simulator.py:68-71 -- enters the
    reference's Riemann solve in double.  The step operator takes it as the left quad {NaN, 0, low word of r, high word of r}
    (include/dhts.h, dhts_macro_step_fwd): the first cell's update must be the one the interface solver (pinned bit for bit by the
    reference's known answers above) gives for the DOUBLE ghost, and differ from the float32-rounded ghost's on some lanes."""
    import torch
    from dhts import ops
    rng = np.random.default_rng(7)
    L, N, dt, dx, um = 2048, 8, 0.05, 1.0, 15.0
    r = rng.uniform(0.02, 0.6, (L, N)).astype(np.float32)
    u = rng.uniform(1.0, 12.0, (L, N)).astype(np.float32)
    ueq64 = um * (1.0 - np.sqrt(np.maximum(r.astype(np.float64), 0.0) + 1e-5))
    ueq = ueq64.astype(np.float32)
    y = (r * (u - ueq)).astype(np.float32)
    src = rng.uniform(0.05, 0.5, L)                                # inflow densities: doubles with all their bits
    src_u = um * (1.0 - np.sqrt(src + 1e-5))
    ghost = np.zeros((L, 2, 4), np.float32)
    ghost[:, 1] = np.stack([r[:, -1], y[:, -1], u[:, -1], ueq[:, -1]], 1)            # (an outflow copy on the right)
    enc, rnd = ghost.copy(), ghost.copy()
    words = src.view(np.float32).reshape(L, 2)                     # little endian: low word, high word
    enc[:, 0, 0], enc[:, 0, 1], enc[:, 0, 2], enc[:, 0, 3] = np.nan, 0.0, words[:, 0], words[:, 1]
    rnd[:, 0] = np.stack([src.astype(np.float32), np.zeros(L, np.float32), src_u.astype(np.float32), src_u.astype(np.float32)], 1)
    desc = ops.macro_desc(L, N, dt, dx, um)
    t = lambda a: torch.tensor(a, device=cuda)
    out_e = ops.macro_step_fwd(desc, t(r), t(y), t(u), t(ueq), t(enc))
    out_r = ops.macro_step_fwd(desc, t(r), t(y), t(u), t(ueq), t(rnd))
    # the two interfaces of cell 0 through the batch solver, in double
    f64 = lambda a: a.astype(np.float64)
    i0 = np.stack([src, np.zeros(L), src_u, src_u, f64(r[:, 0]), f64(y[:, 0]), f64(u[:, 0]), f64(ueq[:, 0]), np.full(L, um)], 1)
    i1 = np.stack([f64(r[:, 0]), f64(y[:, 0]), f64(u[:, 0]), f64(ueq[:, 0]), f64(r[:, 1]), f64(y[:, 1]), f64(u[:, 1]), f64(ueq[:, 1]),
                   np.full(L, um)], 1)
    fl = ops.arz_interface_batch(torch.tensor(np.concatenate([i0, i1]), device=cuda), dt=dt, dx=dx)["flux"].cpu().numpy()
    c = dt / dx
    nr = (f64(r[:, 0]) + (fl[:L, 0] - fl[L:, 0]) * c).astype(np.float32)
    ny = (f64(y[:, 0]) + (fl[:L, 1] - fl[L:, 1]) * c).astype(np.float32)
    assert np.array_equal(out_e[0][:, 0].cpu().numpy(), nr) and np.array_equal(out_e[1][:, 0].cpu().numpy(), ny)
    assert np.array_equal(out_e[0][:, 1:].cpu().numpy(), out_r[0][:, 1:].cpu().numpy())          # nothing else moves
    differs = ((out_e[0][:, 0] != out_r[0][:, 0]) | (out_e[1][:, 0] != out_r[1][:, 0])).float().mean().item()
    print("first cells that differ from the float32-rounded ghost's: %.1f %%" % (100 * differs))        # (a 1-ulp input: it survives the
    assert 0.01 <= differs                                                                               #  float32 store now and then)


@pytest.mark.gpu
@pytest.mark.parametrize("V", [1, 5, 64])
def test_micro_step_under_a_tensor_head_gap_vs_oracle(cuda, oracle, V):
    """dhts_micro_step_fwd_tensor_head: a dMicroLane whose head gap is a float32 tensor (differentiable itscp hybrid episodes) -- the
    reference's detach_vehicle leaves the gap alone, so the HEAD vehicle's IDM and Euler step run in mixed float32 / double arithmetic and
    the followers' in double.  Against the oracle's restatement (pinned bit for bit by the hybrid fixtures with the library switches in):
    next state bit for bit, Jacobian blocks to float32 rounding; the followers equal the analytic operator's, the head does not always."""
    import torch
    from dhts import ops
    rng = np.random.default_rng(300 + V)
    L = 64
    p = np.sort(rng.uniform(0.0, 12.0 * V, (L, V)).astype(np.float32), axis=1)
    p += np.arange(V, dtype=np.float32)[None, :] * 6.0
    v = rng.uniform(0.0, 25.0, (L, V)).astype(np.float32)
    v[::7, -1] = 0.0                                                          # heads at rest at a red light
    prm = np.empty((L, V, 6)); prm[:] = [60.0, 48.0, 54.0, 0.5, 0.1, 5.0]
    head = np.stack([rng.uniform(0.5, 1000.0, L), rng.uniform(-3.0, 3.0, L)], axis=1).astype(np.float32).astype(np.float64)
    head[::5, 0] = np.float32(999.9999)
    dt = 1.0 / 30.0
    desc = ops.micro_desc(L, V, dt)
    tape = torch.empty(ops.micro_step_tape_numel(desc), dtype=torch.float32, device=cuda)
    params_d = torch.tensor(np.ascontiguousarray(prm.transpose(2, 0, 1)), dtype=torch.float64, device=cuda)
    args = (desc, torch.tensor(p, device=cuda), torch.tensor(v, device=cuda), params_d, torch.tensor(head, device=cuda))
    np_h, nv_h = ops.micro_step_fwd(*args, tape=tape, head_tensor=True)
    np_a, nv_a = ops.micro_step_fwd(*args)
    Vp = (V + 63) // 64 * 64
    tp = tape.view(L, 2, Vp, 4).cpu().numpy()
    head_differs = 0
    for l in range(L):
        o = oracle.micro_step_head_tensor(p[l], v[l], prm[l], head[l, 0], head[l, 1], dt)
        assert np.array_equal(np_h[l].cpu().numpy(), o["np"]) and np.array_equal(nv_h[l].cpu().numpy(), o["nv"]), l
        dq = o["dqs"].reshape(V, 2, 4)
        assert rel_max(tp[l, 0, :V], dq[:, 0]) <= 1e-6 and rel_max(tp[l, 1, :V], dq[:, 1]) <= 1e-6
        assert np.array_equal(nv_h[l, :-1].cpu().numpy(), nv_a[l, :-1].cpu().numpy())          # followers: the analytic operator's step
        head_differs += int(nv_h[l, -1].item() != nv_a[l, -1].item())
    print("heads whose new speed differs from the double step's: %d of %d" % (head_differs, L))
    assert head_differs > 0
