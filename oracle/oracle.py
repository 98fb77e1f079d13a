"""ctypes front-end of oracle/libdhts_oracle.so (plain-C restatement, see dhts_oracle.h).

TEST INFRASTRUCTURE ONLY.  Builds the shared object with gcc on first use.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdhts_oracle.so")
_lib = None

OK, ERR_CFL, ERR_COLLISION = 0, 1, 2


def build(force=False):
    src = os.path.join(_HERE, "dhts_oracle.c")
    hdr = os.path.join(_HERE, "dhts_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libdhts_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_idm_acc.restype = C.c_double
        _lib.oracle_idm_acc.argtypes = [C.c_double] * 9 + [C.c_void_p, C.c_void_p]
        _lib.oracle_idm_jac.argtypes = [C.c_double] * 10 + [C.c_void_p] * 3
        _lib.oracle_arz_riemann.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_arz_dLdR.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        _lib.oracle_arz_flux_prime.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        _lib.oracle_arz_from_r_u.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        _lib.oracle_arz_from_r_y.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        _lib.oracle_macro_step.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_void_p] * 8
        _lib.oracle_macro_step_bwd.argtypes = [C.c_int] + [C.c_void_p] * 5
        _lib.oracle_macro_rollout_fwd.argtypes = [C.c_int] * 3 + [C.c_double] * 3 + [C.c_void_p] * 11
        _lib.oracle_macro_rollout_bwd.argtypes = [C.c_int] * 3 + [C.c_double] + [C.c_void_p] * 19
        _lib.oracle_micro_step.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_double] * 3 + [C.c_void_p] * 4
        _lib.oracle_micro_step_f32.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_double] * 3 + [C.c_void_p] * 4
        _lib.oracle_micro_head_mixed.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_float, C.c_double] + [C.c_void_p] * 3
        _lib.oracle_micro_head_mixed.restype = None
        _lib.oracle_micro_step_bwd.argtypes = [C.c_int] + [C.c_void_p] * 5
        _lib.oracle_micro_rollout_fwd.argtypes = [C.c_int] * 3 + [C.c_double] + [C.c_void_p] * 3 + [C.c_double] * 2 + [C.c_void_p] * 5
        _lib.oracle_micro_rollout_bwd.argtypes = [C.c_int] * 3 + [C.c_void_p] * 8
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- ARZ interface ---------------------------------------------------------------------------------
def arz_riemann(L, R, u_max):
    L, R = _f64(L), _f64(R)
    case = C.c_int(0)
    q0 = np.zeros(4)
    sp = np.zeros(2)
    lib().oracle_arz_riemann(_p(L), _p(R), float(u_max), C.addressof(case), _p(q0), _p(sp))
    return case.value, q0, sp


def arz_dLdR(case, q0, L, R, u_max):
    dL = np.zeros((2, 2), np.float32)
    dR = np.zeros((2, 2), np.float32)
    lib().oracle_arz_dLdR(int(case), _p(_f64(q0)), _p(_f64(L)), _p(_f64(R)), float(u_max), _p(dL), _p(dR))
    return dL, dR


def arz_flux_prime(q0, u_max):
    fp = np.zeros((2, 2), np.float32)
    lib().oracle_arz_flux_prime(_p(_f64(q0)), float(u_max), _p(fp))
    return fp


def arz_from_r_u(r, u, u_max):
    y, q = C.c_float(0), C.c_float(0)
    lib().oracle_arz_from_r_u(float(r), float(u), float(u_max), C.addressof(y), C.addressof(q))
    return np.float32(y.value), np.float32(q.value)


def arz_from_r_y(r, y, u_max):
    u, q = C.c_float(0), C.c_float(0)
    lib().oracle_arz_from_r_y(float(r), float(y), float(u_max), C.addressof(u), C.addressof(q))
    return np.float32(u.value), np.float32(q.value)


# ---- macro lane ------------------------------------------------------------------------------------
def macro_step(state, dt, dx, u_max, want_tape=True):
    """state: float32 [4][N+2] = (r, y, u, u_eq) with ghosts.  Returns dict."""
    st = _f32(state)
    N = st.shape[1] - 2
    nr, ny, nu, nq = (np.zeros(N, np.float32) for _ in range(4))
    dqs = np.zeros((N, 3, 2, 2), np.float32) if want_tape else None
    case = np.zeros(N + 1, np.int32)
    speed = np.zeros((N + 1, 2), np.float64)
    ei = C.c_int(-1)
    rc = lib().oracle_macro_step(N, _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), float(dt), float(dx), float(u_max),
                                 _p(nr), _p(ny), _p(nu), _p(nq), _p(dqs), _p(case), _p(speed), C.addressof(ei))
    return dict(rc=rc, nr=nr, ny=ny, nu=nu, nueq=nq, dqs=dqs, case=case, speed=speed, err_index=ei.value)


def macro_step_bwd(dqs, g_nr, g_ny):
    dqs, g_nr, g_ny = _f32(dqs), _f32(g_nr), _f32(g_ny)
    N = g_nr.shape[0]
    g_r, g_y = np.zeros(N + 2, np.float32), np.zeros(N + 2, np.float32)
    lib().oracle_macro_step_bwd(N, _p(dqs), _p(g_nr), _p(g_ny), _p(g_r), _p(g_y))
    return g_r, g_y


def macro_rollout_fwd(r0, u0, ghost_r, ghost_u, T, dt, dx, u_max, want_tape=True, want_hist=False, out=None):
    """out: the dict a previous call of the same shape returned -- its arrays (the tape above all: 48 B per cell-step) are
    written again instead of allocated, so a timing loop does not pay for the first touch of gigabytes of fresh pages on
    every pass (bench.py's cpu_baseline; with hundreds of threads faulting pages of one address space that dominated)."""
    r0, u0, ghost_r, ghost_u = _f32(r0), _f32(u0), _f32(ghost_r), _f32(ghost_u)
    r0 = r0.reshape(-1, r0.shape[-1])
    u0 = u0.reshape(r0.shape)
    L, N = r0.shape
    ghost_r, ghost_u = ghost_r.reshape(L, 2), ghost_u.reshape(L, 2)
    reuse = out is not None and out.get("tape") is not None and out["tape"].shape == (T, L, N, 3, 2, 2) and want_tape and not want_hist
    rT, yT, uT = (out["rT"], out["yT"], out["uT"]) if reuse else (np.zeros((L, N), np.float32) for _ in range(3))
    tape = out["tape"] if reuse else (np.zeros((T, L, N, 3, 2, 2), np.float32) if want_tape else None)
    hr = hy = hu = None
    if want_hist:
        hr, hy, hu = (np.zeros((T, L, N), np.float32) for _ in range(3))
    rc = lib().oracle_macro_rollout_fwd(L, N, T, float(dt), float(dx), float(u_max), _p(r0), _p(u0), _p(ghost_r),
                                        _p(ghost_u), _p(rT), _p(yT), _p(uT), _p(tape), _p(hr), _p(hy), _p(hu))
    return dict(rc=rc, rT=rT, yT=yT, uT=uT, tape=tape, hist_r=hr, hist_y=hy, hist_u=hu,
                r0=r0, u0=u0, ghost_r=ghost_r, ghost_u=ghost_u, T=T, u_max=u_max)


def macro_rollout_bwd(fwd, g_rT=None, g_yT=None, g_uT=None, gh_r=None, gh_y=None, gh_u=None):
    L, N = fwd["r0"].shape
    T = fwd["T"]
    g_rT, g_yT, g_uT, gh_r, gh_y, gh_u = map(_f32, (g_rT, g_yT, g_uT, gh_r, gh_y, gh_u))
    g_r0, g_u0 = np.zeros((L, N), np.float32), np.zeros((L, N), np.float32)
    g_gr, g_gu = np.zeros((L, 2), np.float32), np.zeros((L, 2), np.float32)
    lib().oracle_macro_rollout_bwd(L, N, T, float(fwd["u_max"]), _p(fwd["tape"]), _p(fwd["r0"]), _p(fwd["u0"]),
                                   _p(fwd["ghost_r"]), _p(fwd["ghost_u"]), _p(fwd["rT"]), _p(fwd["yT"]),
                                   _p(g_rT), _p(g_yT), _p(g_uT), _p(fwd["hist_r"]), _p(fwd["hist_y"]),
                                   _p(gh_r), _p(gh_y), _p(gh_u), _p(g_r0), _p(g_u0), _p(g_gr), _p(g_gu))
    return dict(g_r0=g_r0, g_u0=g_u0, g_ghost_r=g_gr, g_ghost_u=g_gu)


# ---- IDM / micro lane ------------------------------------------------------------------------------
def idm_acc(a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt):
    s = C.c_double(0)
    fl = (C.c_int * 2)()
    acc = lib().oracle_idm_acc(a_max, a_pref, v, v_t, dp, dv, s0, Tp, dt, C.addressof(s), C.addressof(fl))
    return acc, s.value, (fl[0], fl[1])


def idm_jac(a_max, a_pref, v, v_t, dp, dv, s0, Tp, sstar, dt, flags):
    fl = (C.c_int * 2)(int(flags[0]), int(flags[1]))
    dE, dLd = np.zeros((2, 2), np.float32), np.zeros((2, 2), np.float32)
    lib().oracle_idm_jac(a_max, a_pref, v, v_t, dp, dv, s0, Tp, sstar, dt, C.addressof(fl), _p(dE), _p(dLd))
    return dE, dLd


def micro_step(p, v, params, head_dp, head_dv, dt, want_tape=True):
    p, v, params = _f32(p), _f32(v), _f64(params)
    V = p.shape[0]
    np_, nv_ = np.zeros(V, np.float32), np.zeros(V, np.float32)
    dqs = np.zeros((V, 2, 2, 2), np.float32) if want_tape else None
    ei = C.c_int(-1)
    rc = lib().oracle_micro_step(V, _p(p), _p(v), _p(params), float(head_dp), float(head_dv), float(dt),
                                 _p(np_), _p(nv_), _p(dqs), C.addressof(ei))
    return dict(rc=rc, np=np_, nv=nv_, dqs=dqs, err_index=ei.value)


def set_numpy_mean(on):
    """The network oracles' running means as numpy computes them on the reference's float32 array (rms.py) -- O(window) per sample."""
    lib().oracle_set_numpy_mean(1 if on else 0)


def set_source_ghost_f64(on):
    """1 (the default): the upstream ghost of an itscp source lane enters the Riemann solve in double, as the reference's Python floats do
    (_simulator.py:68-71); 0: its float32 rounding (oracle and kernels until the end of round 5)."""
    lib().oracle_set_source_ghost_f64(1 if on else 0)


def micro_step_f32(p, v, params, head_dp, head_dv, dt, want_tape=True):
    """The same step in the reference's float32 TENSOR ladder (plain MicroLane on torch tensors: itscp `micro` mode)."""
    p, v, params = _f32(p), _f32(v), _f64(params)
    V = p.shape[0]
    np_, nv_ = np.zeros(V, np.float32), np.zeros(V, np.float32)
    dqs = np.zeros((V, 2, 2, 2), np.float32) if want_tape else None
    ei = C.c_int(-1)
    rc = lib().oracle_micro_step_f32(V, _p(p), _p(v), _p(params), float(head_dp), float(head_dv), float(dt),
                                     _p(np_), _p(nv_), _p(dqs), C.addressof(ei))
    return dict(rc=rc, np=np_, nv=nv_, dqs=dqs, err_index=ei.value)


def micro_step_head_tensor(p, v, params, head_dp, head_dv, dt):
    """A dMicroLane's step under a float32 TENSOR head gap (a differentiable itscp hybrid episode): followers in double (micro_step), the
    head vehicle in the mixed float32 / double arithmetic of the reference (oracle_micro_head_mixed)."""
    p, v, params = _f32(p), _f32(v), _f64(params)
    o = micro_step(p, v, params, float(np.float32(head_dp)), float(np.float32(head_dv)), dt)
    lib().oracle_micro_head_mixed(p.shape[0], _p(p), _p(v), _p(params), float(np.float32(head_dp)), float(np.float32(head_dv)), float(dt),
                                  _p(o["np"]), _p(o["nv"]), _p(o["dqs"]))
    return o


def micro_step_bwd(dqs, g_np, g_nv):
    dqs, g_np, g_nv = _f32(dqs), _f32(g_np), _f32(g_nv)
    V = g_np.shape[0]
    g_p, g_v = np.zeros(V + 1, np.float32), np.zeros(V + 1, np.float32)
    lib().oracle_micro_step_bwd(V, _p(dqs), _p(g_np), _p(g_nv), _p(g_p), _p(g_v))
    return g_p, g_v


def micro_rollout_fwd(p0, v0, params, T, dt, head_dp=1000.0, head_dv=0.0, want_tape=True, want_hist=False, out=None):
    """out: see macro_rollout_fwd (arrays of a previous call of the same shape are reused)."""
    p0, v0, params = _f32(p0), _f32(v0), _f64(params)
    p0 = p0.reshape(-1, p0.shape[-1])
    v0 = v0.reshape(p0.shape)
    L, V = p0.shape
    params = params.reshape(L, V, 6)
    reuse = out is not None and out.get("tape") is not None and out["tape"].shape == (T, L, V, 2, 2, 2) and want_tape and not want_hist
    pT, vT = (out["pT"], out["vT"]) if reuse else (np.zeros((L, V), np.float32), np.zeros((L, V), np.float32))
    tape = out["tape"] if reuse else (np.zeros((T, L, V, 2, 2, 2), np.float32) if want_tape else None)
    hp = hv = None
    if want_hist:
        hp, hv = np.zeros((T, L, V), np.float32), np.zeros((T, L, V), np.float32)
    rc = lib().oracle_micro_rollout_fwd(L, V, T, float(dt), _p(p0), _p(v0), _p(params), float(head_dp), float(head_dv),
                                        _p(pT), _p(vT), _p(tape), _p(hp), _p(hv))
    return dict(rc=rc, pT=pT, vT=vT, tape=tape, hist_p=hp, hist_v=hv, T=T, shape=(L, V))


def micro_rollout_bwd(fwd, g_pT=None, g_vT=None, gh_p=None, gh_v=None):
    L, V = fwd["shape"]
    g_pT, g_vT, gh_p, gh_v = map(_f32, (g_pT, g_vT, gh_p, gh_v))
    g_p0, g_v0 = np.zeros((L, V), np.float32), np.zeros((L, V), np.float32)
    g_head = np.zeros((L, 2), np.float32)
    lib().oracle_micro_rollout_bwd(L, V, fwd["T"], _p(fwd["tape"]), _p(g_pT), _p(g_vT), _p(gh_p), _p(gh_v),
                                   _p(g_p0), _p(g_v0), _p(g_head))
    return dict(g_p0=g_p0, g_v0=g_v0, g_head=g_head)


# ---- macro road network with signals (itscp `macro` mode) ---------------------------------------------------
class NetDesc(C.Structure):
    _fields_ = [("n_lanes", C.c_int), ("n_cells", C.c_int), ("T", C.c_int), ("n_inter_sq", C.c_int),
                ("frames_per_phase", C.c_int), ("n_action", C.c_int), ("dt", C.c_double), ("u_max", C.c_double),
                ("static_speed", C.c_double), ("vehicle_length", C.c_double)]


def net_macro(tab, action, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, want_grad=True, hard=False):
    """tab: dhts.network.MacroNetworkTables (plain numpy tables).  Returns reward, queue [T][L], hist, g_action.
    hard: an evaluation episode (ItscpEnv.step(action, False)): hard thresholds, no gradient."""
    l = lib()
    l.oracle_set_hard(1 if hard else 0)
    want_grad = want_grad and not hard
    l.oracle_net_macro_fwd.argtypes = [C.POINTER(NetDesc)] + [C.c_void_p] * 15
    l.oracle_net_macro_bwd.argtypes = [C.POINTER(NetDesc)] + [C.c_void_p] * 15
    action = _f32(action)
    T, L, Cn = tab.T, tab.n_lanes, tab.n_cells
    d = NetDesc(L, Cn, T, int(n_inter_sq), int(frames_per_phase), len(action), float(dt), float(u_max), float(static_speed),
                float(vehicle_length))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)     # noqa: E731
    ncell, off, dx = i32(tab.lane_ncell), i32(tab.lane_off), _f64(tab.lane_dx)
    kind, inter = i32(tab.sig_kind), i32(tab.inter)
    ls, lg, rs, sched = i32(tab.left_src), i32(tab.left_gate), i32(tab.right_src), _f64(tab.schedule)
    hist = np.zeros((T + 1, 4, Cn), np.float32)
    tape = np.zeros((T, Cn, 12), np.float32)
    kc = np.zeros((T, Cn), np.float32)
    queue = np.zeros((T, L), np.float32)
    reward = C.c_double(0)
    args = [_p(ncell), _p(off), _p(dx), _p(kind), _p(inter), _p(ls), _p(lg), _p(rs), _p(sched), _p(action)]
    rc = l.oracle_net_macro_fwd(C.byref(d), *args, _p(hist), _p(tape), _p(kc), _p(queue), C.addressof(reward))
    l.oracle_set_hard(0)
    out = dict(rc=rc, reward=reward.value, queue=queue, hist=hist, kc=kc)
    if want_grad:
        g = np.zeros(len(action), np.float32)
        l.oracle_net_macro_bwd(C.byref(d), *args, _p(hist), _p(tape), _p(kc), _p(queue), _p(g))
        out["g_action"] = g
    return out


def net_hybrid(tab, routes, route_ptr, action, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0,
               t_cut=None, want_grad=True, want_hist=False, hard=False, vehicle_params=None):
    """tab: dhts.network.HybridNetworkTables; routes [n][stride] int (-1 padded) grouped by first lane + route_ptr [L+1]
    (dhts.network.group_routes).  hard: an evaluation episode (ItscpEnv.step(action, False)): hard thresholds, no gradient."""
    l = lib()
    l.oracle_set_hard(1 if hard else 0)
    want_grad = want_grad and not hard
    l.oracle_set_micro_sources.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    src = np.ascontiguousarray(getattr(tab, "lane_source", np.zeros(tab.n_lanes)), dtype=np.int32)
    draws = None if getattr(tab, "draws", None) is None else _f64(tab.draws)
    if src.any():           # itscp `micro` mode: micro source lanes admit vehicles against the recorded draws
        assert draws is not None, "a network with micro source lanes needs its admission draws (set_micro_sources)"
        l.oracle_set_micro_sources(_p(src), _p(draws), len(draws))
    else:
        l.oracle_set_micro_sources(None, None, 0)
    l.oracle_set_micro_tensor_ladder(1 if getattr(tab, "micro_tensor_ladder", False) else 0)
    l.oracle_set_vehicle_params.argtypes = [C.c_void_p]
    vp = None if vehicle_params is None else _f64(vehicle_params)          # [n_routes][6], rows as the GROUPED routes
    assert vp is None or vp.shape == (np.asarray(routes).shape[0], 6)
    l.oracle_set_vehicle_params(None if vp is None else _p(vp))
    l.oracle_net_hybrid.argtypes = ([C.POINTER(NetDesc)] + [C.c_void_p] * 14 + [C.c_int, C.c_int, C.c_void_p, C.c_int]
                                    + [C.c_void_p] * 8)
    action = _f32(action)
    T, L, Cn = tab.T, tab.n_lanes, tab.n_cells
    d = NetDesc(L, Cn, T, int(n_inter_sq), int(frames_per_phase), len(action), float(dt), float(u_max), float(static_speed),
                float(vehicle_length))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)     # noqa: E731
    routes = i32(routes)
    args = [_p(i32(tab.lane_macro)), _p(_f64(tab.lane_length)), _p(i32(tab.lane_ncell)), _p(i32(tab.lane_off)), _p(_f64(tab.lane_dx)),
            _p(i32(tab.sig_kind)), _p(i32(tab.inter)), _p(i32(tab.left_src)), _p(i32(tab.left_gate)), _p(i32(tab.right_src)),
            _p(i32(tab.conv_next)), _p(_f64(tab.schedule)), _p(routes), _p(i32(route_ptr))]
    queue = np.zeros((T, L), np.float32)
    reward, reward_cut = C.c_double(0), C.c_double(0)
    g = np.zeros(len(action), np.float32)
    nsp, ndep = C.c_int(0), C.c_int(0)
    hist = np.zeros((T + 1, 4, Cn), np.float32) if want_hist else None
    kc = np.zeros((T, Cn), np.float32) if want_hist else None
    rc = l.oracle_net_hybrid(C.byref(d), *args, routes.shape[0], routes.shape[1], _p(action), T if t_cut is None else int(t_cut),
                             _p(queue), C.addressof(reward), C.addressof(reward_cut), _p(g) if want_grad else None,
                             C.addressof(nsp), C.addressof(ndep), _p(hist) if want_hist else None, _p(kc) if want_hist else None)
    l.oracle_set_hard(0)
    draws_used = l.oracle_micro_source_draws_used()
    l.oracle_set_micro_sources(None, None, 0)
    return dict(rc=rc, reward=reward.value, reward_cut=reward_cut.value, queue=queue, g_action=g if want_grad else None,
                n_spawned=nsp.value, n_deposits=ndep.value, hist=hist, kc=kc, draws_used=draws_used)


_SQRTF_CB = None


def set_sqrtf_hook(fn):
    """The glue's float32 square root through `fn(float) -> float` (e.g. torch.sqrt of a 0-dim float32 tensor: the reference's
    environment), or None for sqrtf.  Slow (a Python call per evaluation): for the small fixtures."""
    global _SQRTF_CB
    cb_t = C.CFUNCTYPE(C.c_float, C.c_float)
    cb = cb_t(fn) if fn is not None else C.cast(None, cb_t)
    lib().oracle_set_sqrtf_hook(cb)
    _SQRTF_CB = cb            # keep the trampoline alive while the C side holds it
