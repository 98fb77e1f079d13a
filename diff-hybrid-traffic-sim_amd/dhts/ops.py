"""Device-side operators over libdhts.so: raw calls on torch CUDA tensors + torch.autograd.Functions.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping, the optimiser outside); every
per-step computation of the hot path is a hand-written HIP kernel behind the C ABI (include/dhts.h).

Mirrors the reference operators
    dMacroForwardLayer  road/lane/dmacro_lane.py:234-309
    dMicroForwardLayer  road/lane/dmicro_lane.py:228-298
batched over lanes and fused over time steps.
"""
import ctypes as C
import warnings

import torch

from . import _lib
from ._lib import MacroDesc, MicroDesc, check

EPS = 1e-5  # model/macro/_arz.py:2


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError("%s must be a float32 CUDA tensor (got %s on %s)" % (name, t.dtype, t.device))
    return t.contiguous()


def new_error_record(device):
    """Device-side sticky fault record (dhts_error): int32 [code, step, lane, index]."""
    return torch.zeros(4, dtype=torch.int32, device=device)


class CapacityError(RuntimeError):
    """A fused network episode needed more than one of the kernels' fixed capacities (include/dhts.h: vehicles per micro lane,
    vehicles per episode, records per step).  Nothing was changed on the host: callers can run the episode another way
    (ItscpEnv.step falls back to the lane-by-lane path).  `.index` = the fault record's index field: -2 = the stepwise path's hand-off
    event list (dhts_netstep_tables::max_events), -3 = a reverse sweep whose plan is not its forward's, else 0 / a count."""
    index = 0


def raise_on_fault(err):
    """Read the record back (synchronises) and raise the way the reference asserts."""
    code, step, lane, index = err.tolist()
    if code == _lib.FAULT_CFL:
        # road/lane/_macro_lane.py:145-146
        raise AssertionError("Time step size does not meet CFL condition. Please try smaller delta_time. "
                             "(step %d, lane %d, interface %d)" % (step, lane, index))
    if code == _lib.FAULT_NAN:
        raise AssertionError("non-finite gradient in the reverse sweep (step %d, lane %d)" % (step, lane))   # dmacro_lane.py:308
    if code == _lib.FAULT_CAPACITY:
        e = CapacityError("hybrid network: a fixed capacity was exceeded (record stream / vehicles / lane list / routes / events); "
                          "index %d" % index)
        e.index = int(index)
        raise e
    if code == _lib.FAULT_COLLISION:
        # printed and tolerated in the reference (_micro_lane.py:155-160): the deltas of that vehicle are zeroed, the run goes on
        print("Collision detected between vehicles (step %d, lane %d, vehicle %d)" % (step, lane, index))
    return code


# ---------------------------------------------------------------------------------------------------------
# macro
# ---------------------------------------------------------------------------------------------------------
def macro_desc(L, N, dt, dx, u_max):
    if not (1 <= N <= _lib.MACRO_MAX_CELLS):
        raise ValueError("cells per lane must be in 1..%d" % _lib.MACRO_MAX_CELLS)
    return MacroDesc(int(L), int(N), float(dt), float(dx), float(u_max))


def macro_tape_numel(desc, T):
    """float32 elements of the rollout tape (include/dhts.h: left-cell states + the compacted products of the exceptions)."""
    return _lib.lib().dhts_macro_tape_bytes(C.byref(desc), int(T)) // 4


def macro_step_tape_numel(desc):
    """float32 elements of the single-step operator's tape (the reference's per-cell blocks)."""
    return _lib.lib().dhts_macro_step_tape_bytes(C.byref(desc)) // 4


def macro_step_fwd(desc, r, y, u, ueq, ghost, tape=None, err=None):
    """One step of L lanes through the operator entry point (dqs-layout tape)."""
    r, y, u, ueq, ghost = (_f32c(t, n) for t, n in ((r, "r"), (y, "y"), (u, "u"), (ueq, "ueq"), (ghost, "ghost")))
    out = tuple(torch.empty_like(r) for _ in range(4))
    check(_lib.lib().dhts_macro_step_fwd(C.byref(desc), _ptr(r), _ptr(y), _ptr(u), _ptr(ueq), _ptr(ghost),
                                         _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(tape), _ptr(err), _stream()),
          "dhts_macro_step_fwd")
    return out


def macro_step_bwd(desc, tape, g_r, g_y, err=None):
    g_r, g_y = _f32c(g_r, "g_r"), _f32c(g_y, "g_y")
    out = (torch.empty_like(g_r), torch.empty_like(g_y))
    g_ghost = torch.zeros(desc.n_lanes, 2, 2, dtype=torch.float64, device=g_r.device)
    check(_lib.lib().dhts_macro_step_bwd(C.byref(desc), _ptr(tape), _ptr(g_r), _ptr(g_y), _ptr(out[0]), _ptr(out[1]),
                                         _ptr(g_ghost), _ptr(err), _stream()), "dhts_macro_step_bwd")
    return out[0], out[1], g_ghost


def macro_state_from_ru(r, u, u_max):
    r, u = _f32c(r, "r"), _f32c(u, "u")
    y, q = torch.empty_like(r), torch.empty_like(r)
    check(_lib.lib().dhts_macro_state_from_ru(r.numel(), float(u_max), _ptr(r), _ptr(u), _ptr(y), _ptr(q), _stream()),
          "dhts_macro_state_from_ru")
    return y, q


def macro_state_from_ru_bwd(r, u, g_y, g_r, u_max):
    """g_r is updated in place; returns g_u."""
    g_u = torch.empty_like(r)
    check(_lib.lib().dhts_macro_state_from_ru_bwd(r.numel(), float(u_max), _ptr(r), _ptr(u), _ptr(g_y), _ptr(g_r),
                                                  _ptr(g_u), _stream()), "dhts_macro_state_from_ru_bwd")
    return g_u


def macro_u_tap_bwd(r, y, g_u, g_r, g_y, u_max):
    """g_r, g_y updated in place."""
    check(_lib.lib().dhts_macro_u_tap_bwd(r.numel(), float(u_max), _ptr(r), _ptr(y), _ptr(g_u), _ptr(g_r), _ptr(g_y),
                                          _stream()), "dhts_macro_u_tap_bwd")


def macro_rollout_fwd(desc, T, r, y, u, ueq, ghost, tape=None, hist=None, err=None, out=None):
    """state planes [L][N]; ghost [L][2][4]; returns (r, y, u, ueq) after T steps."""
    L, N = desc.n_lanes, desc.n_cells
    for name, t in (("r", r), ("y", y), ("u", u), ("ueq", ueq)):
        if tuple(t.shape) != (L, N):
            raise ValueError("%s must have shape (%d, %d)" % (name, L, N))
    if tuple(ghost.shape) != (L, 2, 4):
        raise ValueError("ghost must have shape (%d, 2, 4)" % L)
    r, y, u, ueq, ghost = (_f32c(t, n) for t, n in ((r, "r"), (y, "y"), (u, "u"), (ueq, "ueq"), (ghost, "ghost")))
    if out is None:
        out = tuple(torch.empty_like(r) for _ in range(4))
    check(_lib.lib().dhts_macro_rollout_fwd(C.byref(desc), int(T), _ptr(r), _ptr(y), _ptr(u), _ptr(ueq), _ptr(ghost),
                                            _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]),
                                            _ptr(tape), _ptr(hist), _ptr(err), _stream()), "dhts_macro_rollout_fwd")
    return out


def macro_rollout_bwd(desc, T, tape, g_r, g_y, g_hist=None, err=None, out=None, g_ghost=None):
    """returns (g_r0, g_y0, g_ghost[L][2][2] float64)."""
    g_r, g_y = _f32c(g_r, "g_r"), _f32c(g_y, "g_y")
    if out is None:
        out = (torch.empty_like(g_r), torch.empty_like(g_y))
    if g_ghost is None:
        g_ghost = torch.zeros(desc.n_lanes, 2, 2, dtype=torch.float64, device=g_r.device)
    check(_lib.lib().dhts_macro_rollout_bwd(C.byref(desc), int(T), _ptr(tape), _ptr(g_r), _ptr(g_y), _ptr(g_hist),
                                            _ptr(out[0]), _ptr(out[1]), _ptr(g_ghost), _ptr(err), _stream()),
          "dhts_macro_rollout_bwd")
    return out[0], out[1], g_ghost


def macro_rollout_plan(desc, T, want_hist=False):
    """Which kernel instantiations dhts_macro_rollout_fwd / _bwd launch for this shape (include/dhts.h)."""
    plan = (C.c_int32 * 8)()
    check(_lib.lib().dhts_macro_rollout_plan(C.byref(desc), int(T), int(bool(want_hist)), C.byref(plan)), "dhts_macro_rollout_plan")
    # fwd_kernel: 0 = two-phase lane kernel, 1 = one-phase, 2 = two-phase pair kernel
    keys = ("fwd_kernel", "fwd_waves", "fwd_passes", "fwd_full_lane", "bwd_pipelined", "bwd_block", "hist", "fwd_lanes_per_group")
    return dict(zip(keys, list(plan)))


def macro_tape_expand(desc, T, tape):
    """The reference's blocks dqs (dmacro_lane.py:56) of all T steps from a rollout tape: float32 [T][L][3][Np][4]."""
    Np = _lib.lib().dhts_padded(desc.n_cells)
    dqs = torch.empty(int(T), desc.n_lanes, 3, Np, 4, dtype=torch.float32, device=tape.device)
    check(_lib.lib().dhts_macro_tape_expand(C.byref(desc), int(T), _ptr(tape), _ptr(dqs), _stream()), "dhts_macro_tape_expand")
    return dqs


class MacroRollout(torch.autograd.Function):
    """T fused differentiable steps of L independent straight ARZ lanes.

    (r0, u0 [L][N], ghost_r, ghost_u [L][2]) -> (rT, yT, uT [L][N]) (+ hist [T][L][3][N] when asked).
    What example/inverse/macro.py does with one dMacroLane in a RoadNetwork (macro.py:34-68,
    _inverse.py:91-99), for L lanes at once: state set by set_state_vector_u, ghosts by
    set_leftmost_cell / set_rightmost_cell, T x RoadNetwork.forward, state read by get_state_vector.
    """

    @staticmethod
    def forward(ctx, r0, u0, ghost_r, ghost_u, T, dt, dx, u_max, want_hist=False, check_faults=True):
        L, N = r0.shape
        desc = macro_desc(L, N, dt, dx, u_max)
        r0c, u0c = _f32c(r0.detach(), "r0"), _f32c(u0.detach(), "u0")
        gr, gu = _f32c(ghost_r.detach(), "ghost_r"), _f32c(ghost_u.detach(), "ghost_u")
        y0, q0 = macro_state_from_ru(r0c, u0c, u_max)
        gy, gq = macro_state_from_ru(gr, gu, u_max)
        ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()            # [L][2][4]
        need_grad = any(t.requires_grad for t in (r0, u0, ghost_r, ghost_u))
        tape = torch.empty(macro_tape_numel(desc, T), dtype=torch.float32, device=r0.device) if need_grad else None
        hist = torch.empty(T, L, 3, N, dtype=torch.float32, device=r0.device) if want_hist else None
        err = new_error_record(r0.device)
        rT, yT, uT, qT = macro_rollout_fwd(desc, T, r0c, y0, u0c, q0, ghost, tape=tape, hist=hist, err=err)
        if check_faults:
            raise_on_fault(err)
        ctx.desc, ctx.T, ctx.u_max, ctx.tape, ctx.want_hist, ctx.check_faults = desc, T, u_max, tape, want_hist, check_faults
        ctx.save_for_backward(r0c, u0c, gr, gu, gq, rT, yT, hist)
        ctx.mark_non_differentiable(qT)
        if want_hist:
            return rT, yT, uT, qT, hist
        return rT, yT, uT, qT

    @staticmethod
    def backward(ctx, g_rT, g_yT, g_uT, _g_qT, g_hist=None):
        r0, u0, gr, gu, gq, rT, yT, hist = ctx.saved_tensors
        desc, T, um = ctx.desc, ctx.T, ctx.u_max
        L, N = desc.n_lanes, desc.n_cells
        dev = r0.device
        g_r = g_rT.contiguous().clone() if g_rT is not None else torch.zeros(L, N, device=dev)
        g_y = g_yT.contiguous().clone() if g_yT is not None else torch.zeros(L, N, device=dev)
        if g_uT is not None:
            macro_u_tap_bwd(rT, yT, g_uT.contiguous(), g_r, g_y, um)
        gh = None
        if ctx.want_hist and g_hist is not None:
            # per-step taps: (r, y) cotangents directly, u cotangent through the float32 glue of that step's state
            hr, hy = hist[:, :, 0].contiguous(), hist[:, :, 1].contiguous()
            ghr, ghy = g_hist[:, :, 0].contiguous().clone(), g_hist[:, :, 1].contiguous().clone()
            macro_u_tap_bwd(hr, hy, g_hist[:, :, 2].contiguous(), ghr, ghy, um)
            gh = torch.stack([ghr, ghy], dim=2).contiguous()                   # [T][L][2][N]
        err = new_error_record(dev)
        g_r0, g_y0, g_ghost = macro_rollout_bwd(desc, T, ctx.tape, g_r, g_y, g_hist=gh, err=err)
        if ctx.check_faults:             # reading the record back synchronises: off inside HIP-graph capture
            raise_on_fault(err)
        g_u0 = macro_state_from_ru_bwd(r0, u0, g_y0, g_r0, um)
        # ghost (r, y) cotangent sums -> ghost (r, u) leaves, in double (the sum is ill-conditioned)
        rr, uu, qq = gr.double(), gu.double(), gq.double()
        dueq = torch.where(rr < 0, torch.zeros_like(rr), -um * 0.5 / torch.sqrt(rr.clamp_min(0) + EPS))
        g_gr = (g_ghost[..., 0] + g_ghost[..., 1] * ((uu - qq) - rr * dueq)).float()
        g_gu = (g_ghost[..., 1] * rr).float()
        return g_r0, g_u0, g_gr, g_gu, None, None, None, None, None, None


def macro_rollout(r0, u0, ghost_r, ghost_u, T, dt, dx, u_max, want_hist=False, check_faults=True):
    return MacroRollout.apply(r0, u0, ghost_r, ghost_u, int(T), float(dt), float(dx), float(u_max), want_hist,
                              check_faults)


# ---------------------------------------------------------------------------------------------------------
# micro
# ---------------------------------------------------------------------------------------------------------
def micro_desc(L, V, dt):
    if not (1 <= V <= _lib.MICRO_MAX_VEHICLES):
        raise ValueError("vehicle slots per lane must be in 1..%d" % _lib.MICRO_MAX_VEHICLES)
    return MicroDesc(int(L), int(V), float(dt))


def micro_tape_numel(desc, T):
    """float32 elements of the rollout tape (second rows of dEgo / dLeading)."""
    return _lib.lib().dhts_micro_tape_bytes(C.byref(desc), int(T)) // 4


def micro_step_tape_numel(desc):
    """float32 elements of the single-step operator's tape (the reference's dqs)."""
    return _lib.lib().dhts_micro_step_tape_bytes(C.byref(desc)) // 4


def micro_step_fwd(desc, p, v, params, head, count=None, tape=None, err=None, tensor_ladder=False, head_tensor=False):
    """One step of L lanes through the operator entry point (dqs-layout tape).  tensor_ladder: the float32 tensor arithmetic of the
    reference's plain MicroLane (dhts_micro_step_fwd_tensor) instead of the analytic operator's float64 ladder; head_tensor: only the
    lanes' head gaps are float32 tensors there (dMicroLane under a tensor gap: dhts_micro_step_fwd_tensor_head)."""
    p, v = _f32c(p, "p"), _f32c(v, "v")
    out = (torch.empty_like(p), torch.empty_like(v))
    lib = _lib.lib()
    fn = lib.dhts_micro_step_fwd_tensor if tensor_ladder else (lib.dhts_micro_step_fwd_tensor_head if head_tensor else lib.dhts_micro_step_fwd)
    check(fn(C.byref(desc), _ptr(p), _ptr(v), _ptr(count), _ptr(params), _ptr(head),
             _ptr(out[0]), _ptr(out[1]), _ptr(tape), _ptr(err), _stream()), "dhts_micro_step_fwd")
    return out


def micro_rollout_fwd(desc, T, p, v, params, head, count=None, tape=None, hist=None, err=None, out=None):
    """p, v [L][V] float32; params [6][L][V] float64; head [L][2] float64; count [L] int32 or None."""
    L, V = desc.n_lanes, desc.capacity
    p, v = _f32c(p, "p"), _f32c(v, "v")
    if tuple(p.shape) != (L, V) or tuple(v.shape) != (L, V):
        raise ValueError("p, v must have shape (%d, %d)" % (L, V))
    if params.dtype != torch.float64 or tuple(params.shape) != (6, L, V):
        raise ValueError("params must be float64 [6][L][V]")
    if head.dtype != torch.float64 or tuple(head.shape) != (L, 2):
        raise ValueError("head must be float64 [L][2]")
    if count is not None and (count.dtype != torch.int32 or tuple(count.shape) != (L,)):
        raise ValueError("count must be int32 [L]")
    params, head = params.contiguous(), head.contiguous()
    if out is None:
        out = (torch.empty_like(p), torch.empty_like(v))
    check(_lib.lib().dhts_micro_rollout_fwd(C.byref(desc), int(T), _ptr(p), _ptr(v), _ptr(count), _ptr(params), _ptr(head),
                                            _ptr(out[0]), _ptr(out[1]), _ptr(tape), _ptr(hist), _ptr(err), _stream()),
          "dhts_micro_rollout_fwd")
    return out


def micro_rollout_bwd(desc, T, tape, g_p, g_v, count=None, g_hist=None, err=None, out=None, g_head=None):
    g_p, g_v = _f32c(g_p, "g_p"), _f32c(g_v, "g_v")
    if out is None:
        out = (torch.empty_like(g_p), torch.empty_like(g_v))
    if g_head is None:
        g_head = torch.zeros(desc.n_lanes, 2, dtype=torch.float64, device=g_p.device)
    check(_lib.lib().dhts_micro_rollout_bwd(C.byref(desc), int(T), _ptr(tape), _ptr(count), _ptr(g_p), _ptr(g_v),
                                            _ptr(g_hist), _ptr(out[0]), _ptr(out[1]), _ptr(g_head), _ptr(err), _stream()),
          "dhts_micro_rollout_bwd")
    return out[0], out[1], g_head


def micro_rollout_plan(desc, T, has_count=False):
    """Which kernel instantiations dhts_micro_rollout_fwd / _bwd launch for this shape (include/dhts.h)."""
    plan = (C.c_int32 * 8)()
    check(_lib.lib().dhts_micro_rollout_plan(C.byref(desc), int(T), int(bool(has_count)), C.byref(plan)), "dhts_micro_rollout_plan")
    keys = ("fwd_waves", "fwd_passes", "fwd_full_lane", "bwd_one_vehicle_per_thread", "bwd_block")
    return dict(zip(keys, list(plan)[:5]))


def micro_step_bwd(desc, tape, g_p, g_v, count=None):
    """dMicroForwardLayer.backward for a batch of lanes: returns (g_p[L][V], g_v[L][V], g_virtual[L][2] float64),
    g_virtual = raw cotangent of the virtual leader slot (not folded into the head vehicle)."""
    g_p, g_v = _f32c(g_p, "g_p"), _f32c(g_v, "g_v")
    out = (torch.empty_like(g_p), torch.empty_like(g_v))
    g_virtual = torch.zeros(desc.n_lanes, 2, dtype=torch.float64, device=g_p.device)
    check(_lib.lib().dhts_micro_step_bwd(C.byref(desc), _ptr(tape), _ptr(count), _ptr(g_p), _ptr(g_v), _ptr(out[0]), _ptr(out[1]),
                                         _ptr(g_virtual), None, _stream()), "dhts_micro_step_bwd")
    return out[0], out[1], g_virtual


class MicroRollout(torch.autograd.Function):
    """T fused differentiable IDM steps of L independent lanes with a fixed head gap.

    (p0, v0 [L][V]) -> (pT, vT) (+ hist [T][L][2][V]).  What example/inverse/micro.py does with one dMicroLane
    (micro.py:36-118): vehicles ordered tail -> head, head gap = lane defaults (1000, 0).
    The head gap is differentiable: `head` [L][2] float64 receives the cotangent of
    (head_position_delta, head_speed_delta).
    """

    @staticmethod
    def forward(ctx, p0, v0, params, head, count, T, dt, want_hist=False, check_faults=True):
        L, V = p0.shape
        desc = micro_desc(L, V, dt)
        p0c, v0c = _f32c(p0.detach(), "p0"), _f32c(v0.detach(), "v0")
        need_grad = p0.requires_grad or v0.requires_grad or head.requires_grad
        tape = torch.empty(micro_tape_numel(desc, T), dtype=torch.float32, device=p0.device) if need_grad else None
        hist = torch.empty(T, L, 2, V, dtype=torch.float32, device=p0.device) if want_hist else None
        err = new_error_record(p0.device)
        pT, vT = micro_rollout_fwd(desc, T, p0c, v0c, params, head.detach(), count=count, tape=tape, hist=hist, err=err)
        if check_faults:                 # a collision is printed like the reference does, and tolerated
            raise_on_fault(err)
        ctx.desc, ctx.T, ctx.tape, ctx.count, ctx.want_hist, ctx.check_faults = desc, T, tape, count, want_hist, check_faults
        # the reverse sweep's own record, made here (no allocation inside backward, i.e. inside a graph capture of it)
        ctx.err_bwd = new_error_record(p0.device) if need_grad else None
        if want_hist:
            return pT, vT, hist
        return pT, vT

    @staticmethod
    def backward(ctx, g_pT, g_vT, g_hist=None):
        desc, T = ctx.desc, ctx.T
        dev = ctx.tape.device
        L, V = desc.n_lanes, desc.capacity
        g_p = g_pT.contiguous() if g_pT is not None else torch.zeros(L, V, device=dev)
        g_v = g_vT.contiguous() if g_vT is not None else torch.zeros(L, V, device=dev)
        gh = g_hist.contiguous() if (ctx.want_hist and g_hist is not None) else None
        err = ctx.err_bwd
        err.zero_()                      # (first fault wins: a second backward with retain_graph must not report the first one's)
        g_p0, g_v0, g_head = micro_rollout_bwd(desc, T, ctx.tape, g_p, g_v, count=ctx.count, g_hist=gh, err=err)
        # The record only feeds a warning (the reference's micro backward returns NaNs silently, dmicro_lane.py:271-298): it is not
        # read back here -- that would be a host synchronisation per reverse sweep -- unless the gradient that is being returned
        # anyway is non-finite, which the caller's next use of it would synchronise on as well.
        if ctx.check_faults and not torch.cuda.is_current_stream_capturing():
            MicroRollout.last_bwd_record = err           # (dhts.ops.micro_bwd_fault() reads it on demand)
        return g_p0, g_v0, None, g_head, None, None, None, None, None


def micro_bwd_fault(warn=True):
    """Where the most recent micro reverse sweep first met a non-finite cotangent: (step, lane, vehicle) or None.  Reads the sweep's
    fault record back (a host synchronisation: call it once per so many iterations, or when a gradient looks wrong)."""
    rec = getattr(MicroRollout, "last_bwd_record", None)
    if rec is None:
        return None
    code, step, lane, index = rec.tolist()
    if code != _lib.FAULT_NAN:
        return None
    if warn:
        warnings.warn("non-finite gradient in the micro reverse sweep (step %d, lane %d, vehicle %d)" % (step, lane, index), RuntimeWarning)
    return step, lane, index


def micro_rollout(p0, v0, params, head, T, dt, count=None, want_hist=False, check_faults=True):
    return MicroRollout.apply(p0, v0, params, head, count, int(T), float(dt), want_hist, bool(check_faults))


# ---------------------------------------------------------------------------------------------------------
# known-answer / scalar-surface entry points (model.macro._arz, model.macro.darz, model.micro._idm mirrors)
# ---------------------------------------------------------------------------------------------------------
def arz_interface_batch(inp, dt=0.01, dx=5.0, variant=0):
    """inp: float64 [n][9] = rL yL uL ueqL rR yR uR ueqR u_max (CUDA).  Returns a dict of CUDA tensors:
    case [n] int32, q0 [n][4] f64, flux [n][2] f64, dL/dR/fp/A/B [n][2][2] f32, cfl_bad [n] bool, speed [n][2] f64
    (speed0, speed1 of ARZ.riemann_solve)."""
    if inp.dtype != torch.float64 or not inp.is_cuda or inp.dim() != 2 or inp.shape[1] != 9:
        raise TypeError("inp must be a float64 CUDA tensor of shape [n][9]")
    n, dev = inp.shape[0], inp.device
    soa = inp.t().contiguous()
    case = torch.empty(n, dtype=torch.int32, device=dev)
    q0 = torch.empty(4, n, dtype=torch.float64, device=dev)
    flux = torch.empty(2, n, dtype=torch.float64, device=dev)
    f32 = [torch.empty(4, n, dtype=torch.float32, device=dev) for _ in range(5)]
    bad = torch.empty(n, dtype=torch.int32, device=dev)
    speed = torch.empty(2, n, dtype=torch.float64, device=dev)
    check(_lib.lib().dhts_arz_interface_batch(n, int(variant), _ptr(soa), float(dt), float(dx), _ptr(case), _ptr(q0), _ptr(flux),
                                              *[_ptr(t) for t in f32], _ptr(bad), _ptr(speed), _stream()), "dhts_arz_interface_batch")
    m = [t.t().reshape(n, 2, 2) for t in f32]
    return dict(case=case, q0=q0.t().contiguous(), flux=flux.t().contiguous(), dL=m[0], dR=m[1], fp=m[2], A=m[3], B=m[4],
                cfl_bad=bad.bool(), speed=speed.t().contiguous())


def idm_batch(inp, variant=0):
    """inp: float64 [n][9] = a_max a_pref v v_target dp dv min_space time_pref dt (CUDA).
    Returns next_v (float32-rounded v + dt acc, as f64), acc, dEgo, dLeading [n][2][2] f32, collided [n] bool."""
    if inp.dtype != torch.float64 or not inp.is_cuda or inp.dim() != 2 or inp.shape[1] != 9:
        raise TypeError("inp must be a float64 CUDA tensor of shape [n][9]")
    n, dev = inp.shape[0], inp.device
    soa = inp.t().contiguous()
    nxt = torch.empty(2, n, dtype=torch.float64, device=dev)
    dE = torch.empty(4, n, dtype=torch.float32, device=dev)
    dLd = torch.empty(4, n, dtype=torch.float32, device=dev)
    col = torch.empty(n, dtype=torch.int32, device=dev)
    acs = torch.empty(2, n, dtype=torch.float64, device=dev)
    clips = torch.empty(2, n, dtype=torch.int32, device=dev)
    check(_lib.lib().dhts_idm_batch(n, int(variant), _ptr(soa), _ptr(nxt), _ptr(dE), _ptr(dLd), _ptr(col), _ptr(acs), _ptr(clips), _stream()),
          "dhts_idm_batch")
    return dict(next_p=nxt[0], next_v=nxt[1], dEgo=dE.t().reshape(n, 2, 2), dLeading=dLd.t().reshape(n, 2, 2),
                collided=col.bool(), acc=acs[0], sstar=acs[1], clipped_acc=clips[0].bool(), clipped_spacing=clips[1].bool())


def idm_jac_batch(inp):
    """inp: float64 [n][12] = a_max a_pref v v_target dp dv min_space time_pref optimal_spacing dt clipped_acceleration
    clipped_optimal_spacing (CUDA) -> dEgo, dLeading [n][2][2] float32: dIDM.compute_dEgo / compute_dLeading with the CALLER's
    optimal spacing and clip flags (reference didm.py:13-103)."""
    if inp.dtype != torch.float64 or not inp.is_cuda or inp.dim() != 2 or inp.shape[1] != 12:
        raise TypeError("inp must be a float64 CUDA tensor of shape [n][12]")
    n, dev = inp.shape[0], inp.device
    soa = inp.t().contiguous()
    dE = torch.empty(4, n, dtype=torch.float32, device=dev)
    dLd = torch.empty(4, n, dtype=torch.float32, device=dev)
    check(_lib.lib().dhts_idm_jac_batch(n, _ptr(soa), _ptr(dE), _ptr(dLd), _stream()), "dhts_idm_jac_batch")
    return dE.t().reshape(n, 2, 2), dLd.t().reshape(n, 2, 2)


# ---------------------------------------------------------------------------------------------------------
# macro road network with differentiable signals (itscp `macro` mode), replica batch
# ---------------------------------------------------------------------------------------------------------
class DeviceNetTables:
    """dhts.network.MacroNetworkTables uploaded once; `tables` may be one MacroNetworkTables (shared by all replicas)
    or a list of them (one per replica: same topology, own schedules / per-step routes)."""

    def __init__(self, tables, device):
        import numpy as np
        many = isinstance(tables, (list, tuple))
        first = tables[0] if many else tables
        self.n_lanes, self.n_cells, self.T = first.n_lanes, first.n_cells, first.T
        self.n_replica_tables = len(tables) if many else 0
        up = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)    # noqa: E731
        self.lane_ncell, self.lane_off = up(first.lane_ncell, torch.int32), up(first.lane_off, torch.int32)
        self.sig_kind, self.inter = up(first.sig_kind, torch.int32), up(first.inter, torch.int32)
        self.lane_dx = up(first.lane_dx, torch.float64)
        stack = (lambda name: np.stack([getattr(t, name) for t in tables])) if many else (lambda name: getattr(first, name))
        self.left_src, self.left_gate = up(stack("left_src"), torch.int32), up(stack("left_gate"), torch.int32)
        self.right_src, self.schedule = up(stack("right_src"), torch.int32), up(stack("schedule"), torch.float64)
        pad1 = lambda a: a if len(a) else np.zeros(1, dtype=np.int32)      # noqa: E731  (never pass a NULL pointer)
        self.nxt_ptr, self.nxt_idx = up(first.nxt_ptr, torch.int32), up(pad1(first.nxt_idx), torch.int32)
        self.prv_ptr, self.prv_idx = up(first.prv_ptr, torch.int32), up(pad1(first.prv_idx), torch.int32)
        self.c = _lib.NetTables(self.lane_ncell.data_ptr(), self.lane_off.data_ptr(), self.sig_kind.data_ptr(),
                                self.inter.data_ptr(), self.lane_dx.data_ptr(), self.left_src.data_ptr(),
                                self.left_gate.data_ptr(), self.right_src.data_ptr(), self.schedule.data_ptr(),
                                self.T * self.n_lanes if many else 0, self.nxt_ptr.data_ptr(), self.nxt_idx.data_ptr(),
                                self.prv_ptr.data_ptr(), self.prv_idx.data_ptr(), first.n_edges)


class NetMacroRollout(torch.autograd.Function):
    """action [R][A] -> reward [R] of R replicas of a signalised macro network (ItscpEnv.step(action, True) of the
    reference in `macro` mode, reward = - sum of squared queue lengths); also returns the per-step queue terms."""

    @staticmethod
    def forward(ctx, action, dev_tables, n_inter_sq, frames_per_phase, dt, u_max, static_speed, vehicle_length, check_faults=True,
                err=None):
        a = _f32c(action.detach(), "action")
        R, A = a.shape
        if dev_tables.n_replica_tables not in (0, R):
            raise ValueError("per-replica tables must match the number of replicas")
        d = _lib.NetDesc(R, dev_tables.n_lanes, dev_tables.n_cells, dev_tables.T, int(n_inter_sq), int(frames_per_phase), A,
                         float(dt), float(u_max), float(static_speed), float(vehicle_length))
        lib = _lib.lib()
        hist_n, tape_n = lib.dhts_net_macro_hist_bytes(C.byref(d)) // 4, lib.dhts_net_macro_tape_bytes(C.byref(d)) // 4
        if hist_n == 0:
            raise ValueError("unsupported network size (need cells + lanes <= 1024 and lanes <= cells)")
        dev = a.device
        hist = torch.empty(hist_n, dtype=torch.float32, device=dev)
        tape = torch.empty(tape_n, dtype=torch.float32, device=dev)
        kc = torch.empty(R, dev_tables.T, dev_tables.n_cells, dtype=torch.float32, device=dev)
        queue = torch.empty(R, dev_tables.T, dev_tables.n_lanes, dtype=torch.float32, device=dev)
        reward = torch.empty(R, dtype=torch.float32, device=dev)
        ws = torch.zeros(R * dev_tables.T * 2 * dev_tables.n_lanes, dtype=torch.float32, device=dev)
        own_err = err is None              # a caller's record is sticky across calls and read by the caller (no sync here)
        if own_err:
            err = new_error_record(dev)
        check(lib.dhts_net_macro_rollout_fwd(C.byref(d), C.byref(dev_tables.c), _ptr(a), _ptr(hist), _ptr(tape), _ptr(kc),
                                             _ptr(queue), _ptr(reward), _ptr(ws), _ptr(err), _stream()),
              "dhts_net_macro_rollout_fwd")
        if check_faults and own_err:     # reading the record back synchronises: off inside HIP-graph capture
            raise_on_fault(err)
        ctx.d, ctx.tables, ctx.check_faults = d, dev_tables, bool(check_faults) and own_err
        ctx.err = None if own_err else err
        ctx.save_for_backward(a, hist, tape, kc, queue, ws)
        ctx.mark_non_differentiable(queue)
        return reward, queue

    @staticmethod
    def backward(ctx, g_reward, _g_queue):
        a, hist, tape, kc, queue, ws = ctx.saved_tensors
        d = ctx.d
        g_action = torch.empty_like(a)
        err = ctx.err if ctx.err is not None else new_error_record(a.device)
        g = g_reward.contiguous().float()          # a named local: the (possibly fresh) tensor must outlive the launch
        check(_lib.lib().dhts_net_macro_rollout_bwd(C.byref(d), C.byref(ctx.tables.c), _ptr(a), _ptr(hist), _ptr(tape), _ptr(kc),
                                                    _ptr(queue), _ptr(g), _ptr(g_action), _ptr(ws), _ptr(err),
                                                    _stream()), "dhts_net_macro_rollout_bwd")
        if ctx.check_faults:
            raise_on_fault(err)
        return g_action, None, None, None, None, None, None, None, None, None


def net_macro_rollout(action, dev_tables, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0,
                      check_faults=True, err=None):
    """Returns (reward [R], queue [R][T][L]).  check_faults=False: nothing is read back (no host sync; usable inside a
    HIP-graph capture); faults stay in the device record.  err: a caller-owned sticky fault record (new_error_record) used
    by both directions and read by the caller when it wants to (raise_on_fault)."""
    return NetMacroRollout.apply(action, dev_tables, int(n_inter_sq), int(frames_per_phase), float(dt), float(u_max),
                                 float(static_speed), float(vehicle_length), bool(check_faults), err)


def net_macro_eval(action, dev_tables, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, err=None):
    """An evaluation episode (ItscpEnv.step(action, False) of the reference: hard signal / boundary / is_static thresholds,
    trainer.py:94-142) of R replicas of a macro network: returns (reward [R], queue [R][T][L]); nothing differentiable."""
    a = _f32c(action.detach(), "action")
    R, A = a.shape
    if dev_tables.n_replica_tables not in (0, R):
        raise ValueError("per-replica tables must match the number of replicas")
    d = _lib.NetDesc(R, dev_tables.n_lanes, dev_tables.n_cells, dev_tables.T, int(n_inter_sq), int(frames_per_phase), A,
                     float(dt), float(u_max), float(static_speed), float(vehicle_length))
    queue = torch.empty(R, dev_tables.T, dev_tables.n_lanes, dtype=torch.float32, device=a.device)
    reward = torch.empty(R, dtype=torch.float32, device=a.device)
    own_err = err is None
    if own_err:
        err = new_error_record(a.device)
    check(_lib.lib().dhts_net_macro_rollout_eval(C.byref(d), C.byref(dev_tables.c), _ptr(a), _ptr(queue), _ptr(reward), _ptr(err),
                                                 _stream()), "dhts_net_macro_rollout_eval")
    if own_err:
        raise_on_fault(err)
    return reward, queue


class DeviceHybridTables:
    """dhts.network.HybridNetworkTables plus the pre-drawn vehicle routes [n_routes][stride] (int, -1 padded; the k-th
    vehicle spawned onto a lane takes the k-th route starting there, cyclically), uploaded once.  `tables` may be one
    HybridNetworkTables (shared by all replicas) or a list of them (one per replica: same topology, own inflow schedules
    and per-step macro routes)."""

    def __init__(self, tables, routes, device, records_per_step=0, lane_capacity=0, vehicle_params=None):
        """vehicle_params [n_routes][6] (rows as `routes`): the IDM attributes of the vehicle that takes each route row
        (dhts_hybrid_tables::veh_params); None = every vehicle a default_micro_vehicle(speed_limit)."""
        import numpy as np
        if lane_capacity not in (0, 16, 32, 64, 128):
            raise ValueError("lane_capacity (vehicles a micro lane holds at once) must be 0 (= 16), 16, 32, 64 or 128")
        self.lane_capacity = int(lane_capacity)
        self.two_per_cu = 0                # dhts_hybrid_tables::two_per_cu: 0 = DHTS_OPT_HYB_PACK decides, 1 = packed, -1 = one replica per unit
        many = isinstance(tables, (list, tuple))
        t = tables[0] if many else tables
        for i, x in enumerate(tables if many else [t]):
            x.check_kernel_limits()
            if (x.n_lanes, x.n_cells, x.T) != (t.n_lanes, t.n_cells, t.T) or not np.array_equal(np.asarray(x.lane_source), np.asarray(t.lane_source)):
                raise ValueError("per-replica tables must share the topology of table 0 (lanes, cells, steps, source lanes): table %d differs" % i)
            if np.asarray(x.lane_source).any() and getattr(x, "draws", None) is None:
                raise ValueError("table %d has micro source lanes but no admission draws (HybridNetworkTables.set_micro_sources)" % i)
        self.n_lanes, self.n_cells, self.T = t.n_lanes, t.n_cells, t.T
        self.n_replica_tables = len(tables) if many else 0
        up = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)    # noqa: E731
        pad1 = lambda a: a if len(a) else np.zeros(1, dtype=np.int32)      # noqa: E731
        stack = (lambda name: np.stack([getattr(x, name) for x in tables])) if many else (lambda name: getattr(t, name))
        from .network import group_routes
        routes = np.ascontiguousarray(routes, dtype=np.int32)
        if routes.ndim != 2 or routes.shape[0] < 1 or routes.shape[1] > 32:
            raise ValueError("routes must be [n_routes >= 1][stride <= 32]")
        self._veh_params = None
        if vehicle_params is not None:
            routes, route_ptr, vp = group_routes(routes, t.n_lanes, vehicle_params)
            self._veh_params = up(vp, torch.float64)
        else:
            routes, route_ptr = group_routes(routes, t.n_lanes)
        self.n_routes, self.route_stride = int(routes.shape[0]), int(routes.shape[1])
        self.records_per_step = int(records_per_step)
        self.n_micro = int((np.asarray(t.lane_macro) == 0).sum())
        self._keep = [up(t.lane_ncell, torch.int32), up(t.lane_off, torch.int32), up(t.sig_kind, torch.int32), up(t.inter, torch.int32),
                      up(t.lane_dx, torch.float64), up(stack("left_src"), torch.int32), up(stack("left_gate"), torch.int32),
                      up(stack("right_src"), torch.int32), up(stack("schedule"), torch.float64), up(t.nxt_ptr, torch.int32),
                      up(pad1(t.nxt_idx), torch.int32), up(t.prv_ptr, torch.int32), up(pad1(t.prv_idx), torch.int32),
                      up(t.lane_macro, torch.int32), up(t.lane_length, torch.float64), up(stack("conv_next"), torch.int32),
                      up(routes, torch.int32), up(route_ptr, torch.int32)]
        # micro source lanes (itscp `micro` mode): the lane flags and the host's admission draws (per replica when `tables` is a list)
        self.has_sources = bool(np.asarray(t.lane_source).any())
        self.micro_tensor_ladder = bool(getattr(t, "micro_tensor_ladder", False))
        self.n_draws, self.draws_stride = 0, 0
        if self.has_sources:
            if many:
                n = max(len(x.draws) for x in tables)
                d = np.full((len(tables), n), 2.0)           # (a draw of 2.0 admits nobody)
                for i, x in enumerate(tables):
                    d[i, :len(x.draws)] = x.draws
                self.n_draws, self.draws_stride = n, n
            else:
                d = np.asarray(t.draws, dtype=np.float64)
                self.n_draws = len(d)
            self._keep += [up(t.lane_source, torch.int32), up(d, torch.float64)]
        k = [x.data_ptr() for x in self._keep]
        self.net = _lib.NetTables(k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], self.T * self.n_lanes if many else 0,
                                  k[9], k[10], k[11], k[12], t.n_edges)

    def first(self, n_replicas):
        """The same uploaded tables as a batch of the first `n_replicas` replicas (no copy: the per-replica arrays are
        replica-major, so a shorter batch reads a prefix of them)."""
        import copy
        if not 1 <= int(n_replicas) <= max(self.n_replica_tables, 1) and self.n_replica_tables:
            raise ValueError("tables hold %d replicas" % self.n_replica_tables)
        v = copy.copy(self)
        if self.n_replica_tables:
            v.n_replica_tables = int(n_replicas)
        return v

    def set_draws(self, draws):
        """A fresh stream of admission draws for the next episode (micro source lanes; same length as the uploaded one)."""
        import numpy as np
        if not self.has_sources:
            raise ValueError("the network has no micro source lanes")
        d = torch.as_tensor(np.ascontiguousarray(draws, dtype=np.float64), device=self._keep[19].device)
        if d.shape != self._keep[19].shape:
            raise ValueError("draws must keep their shape %s" % (tuple(self._keep[19].shape),))
        self._keep[19].copy_(d)

    def c(self, loss_steps=0):
        k = [x.data_ptr() for x in self._keep]
        src = (k[18], k[19]) if self.has_sources else (None, None)
        return _lib.HybridTables(self.net, k[13], k[14], k[15], k[16], k[17], self.n_routes, self.route_stride, self.records_per_step,
                                 int(loss_steps), self.n_micro, src[0], src[1], self.n_draws, self.draws_stride, self.lane_capacity,
                                 1 if self.micro_tensor_ladder else 0, None if self._veh_params is None else self._veh_params.data_ptr(),
                                 int(self.two_per_cu))


class NetHybridRollout(torch.autograd.Function):
    """action [R][A] -> reward [R] of R replicas of a signalised hybrid network (ItscpEnv.step(action, True) of the
    reference in `hybrid` mode); also returns the per-step queue terms and (spawned, deposited, records) per replica.
    loss_steps > 0 restricts the differentiated reward to the first loss_steps steps (second output `reward_cut`)."""

    @staticmethod
    def forward(ctx, action, dev_tables, n_inter_sq, frames_per_phase, dt, u_max, static_speed, vehicle_length, loss_steps,
                check_faults=True, err=None, err_bwd=None):
        a = _f32c(action.detach(), "action")
        R, A = a.shape
        t = dev_tables
        if t.n_replica_tables not in (0, R):
            raise ValueError("per-replica tables must match the number of replicas")
        d = _lib.NetDesc(R, t.n_lanes, t.n_cells, t.T, int(n_inter_sq), int(frames_per_phase), A, float(dt), float(u_max),
                         float(static_speed), float(vehicle_length))
        tc = t.c(loss_steps)
        lib = _lib.lib()
        ws_n = lib.dhts_net_hybrid_workspace_bytes(C.byref(d), C.byref(tc))
        if ws_n == 0:
            raise ValueError("unsupported hybrid network size")
        dev = a.device
        # (an all-micro network has no cells: the kernels' unconditional prefetches still want something to read)
        hist = torch.empty(max(R * (t.T + 1) * 4 * t.n_cells, 64), dtype=torch.float32, device=dev)
        tape = torch.empty(max(lib.dhts_net_hybrid_tape_bytes(C.byref(d)) // 4, 64), dtype=torch.float32, device=dev)
        kc = torch.empty(max(R * t.T * t.n_cells, 64), dtype=torch.float32, device=dev)
        queue = torch.empty(R, t.T, t.n_lanes, dtype=torch.float32, device=dev)
        reward = torch.empty(R, dtype=torch.float32, device=dev)
        counts = torch.zeros(R, 4, dtype=torch.int32, device=dev)
        ws = torch.empty(ws_n, dtype=torch.uint8, device=dev)
        own_err = err is None              # a caller's record is sticky across calls and read by the caller (no sync here)
        if own_err:
            err = new_error_record(dev)
        check(lib.dhts_net_hybrid_rollout_fwd(C.byref(d), C.byref(tc), _ptr(a), _ptr(hist), _ptr(tape), _ptr(kc), _ptr(queue),
                                              _ptr(reward), _ptr(counts), _ptr(ws), _ptr(err), _stream()),
              "dhts_net_hybrid_rollout_fwd")
        if check_faults and own_err:     # reading the record back synchronises: off inside HIP-graph capture
            try:
                raise_on_fault(err)
            except CapacityError:
                # two replicas per compute unit (more replicas than units) halve the record staging area: the same batch once more
                # with one replica per unit before the caller hears of it (the reverse sweep follows the tables it is given)
                plan = (C.c_int32 * 8)()
                check(lib.dhts_net_hybrid_plan(C.byref(d), C.byref(tc), plan), "dhts_net_hybrid_plan")
                if not plan[0]:
                    raise
                import copy
                t = copy.copy(t)
                t.two_per_cu = -1
                tc = t.c(loss_steps)
                err.zero_()
                counts.zero_()
                check(lib.dhts_net_hybrid_rollout_fwd(C.byref(d), C.byref(tc), _ptr(a), _ptr(hist), _ptr(tape), _ptr(kc), _ptr(queue),
                                                      _ptr(reward), _ptr(counts), _ptr(ws), _ptr(err), _stream()),
                      "dhts_net_hybrid_rollout_fwd")
                raise_on_fault(err)
        # the reverse sweep raises only on a record of its own: a caller-owned one (err, or err_bwd for the reverse sweep alone) is
        # the caller's to read -- that is how a batch tolerates one member's NaN
        ctx.d, ctx.tables, ctx.loss_steps = d, t, int(loss_steps)
        ctx.check_faults = bool(check_faults) and own_err and err_bwd is None
        ctx.err = err_bwd if err_bwd is not None else (None if own_err else err)
        ctx.save_for_backward(a, hist, tape, kc, queue, ws)
        ctx.mark_non_differentiable(reward, queue, counts)
        if loss_steps and loss_steps > 0:
            # reference accumulation order: lanes outer, steps inner
            cut = -(queue[:, :int(loss_steps), :].transpose(1, 2).reshape(R, -1).sum(dim=1))
        else:
            cut = reward.clone()
        return cut, reward, queue, counts

    @staticmethod
    def backward(ctx, g_cut, _g_reward, _g_queue, _g_counts):
        a, hist, tape, kc, queue, ws = ctx.saved_tensors
        d = ctx.d
        tc = ctx.tables.c(ctx.loss_steps)
        g_action = torch.empty_like(a)
        err = ctx.err if ctx.err is not None else new_error_record(a.device)
        g = g_cut.contiguous().float()             # a named local: the (possibly fresh) tensor must outlive the launch
        check(_lib.lib().dhts_net_hybrid_rollout_bwd(C.byref(d), C.byref(tc), _ptr(a), _ptr(hist), _ptr(tape), _ptr(kc), _ptr(queue),
                                                     _ptr(g), _ptr(g_action), _ptr(ws), _ptr(err), _stream()),
              "dhts_net_hybrid_rollout_bwd")
        if ctx.check_faults:
            raise_on_fault(err)     # a NaN in the reverse sweep asserts like the reference (dmacro_lane.py:308)
        return g_action, None, None, None, None, None, None, None, None, None, None, None


class NetHybridStateRollout(torch.autograd.Function):
    """R replicas of a road network of ARZ and IDM lanes that starts from a GIVEN state, all T steps in one launch each way:
    (r0, u0 [R][C]) -> (rT, yT, uT [R][C], veh [R][128][4], events, counts).  With dev_tables built by
    HybridNetworkTables.plain(...) and plain = True this is T x RoadNetwork.forward(dt, True) of the reference on the network
    of example/inverse/hybrid.py (macro -> micro -> macro: flux-capacitor spawns, IDM steps, deposits; hybrid.py:37-146,
    _inverse.py:91-99) with its backward pass: cotangents of the final (r, u) of every cell and of the final (position, speed)
    of every vehicle go back to (r0, u0).  ghost0 [R][L][4] = stored (r, u) of each lane's upstream / downstream ghost
    (set_leftmost_cell / set_rightmost_cell), constants."""

    @staticmethod
    def forward(ctx, r0, u0, ghost0, dev_tables, dt, u_max, plain, vehicle_length, check_faults):
        t = dev_tables
        R, Cc = r0.shape
        if Cc != t.n_cells:
            raise ValueError("state must be [R][%d cells]" % t.n_cells)
        if t.n_replica_tables not in (0, R):
            raise ValueError("per-replica tables must match the number of replicas")
        dev = r0.device
        r0c, u0c = _f32c(r0.detach(), "r0"), _f32c(u0.detach(), "u0")
        y0, q0 = macro_state_from_ru(r0c, u0c, u_max)
        state0 = torch.stack([r0c, y0, u0c, q0], dim=1).contiguous()                       # [R][4][C]
        g0 = None if ghost0 is None else _f32c(ghost0.detach(), "ghost0")
        if g0 is not None and tuple(g0.shape) != (R, t.n_lanes, 4):
            raise ValueError("ghost0 must be [R][%d lanes][4]" % t.n_lanes)
        action = torch.full((R, 1), 0.5, dtype=torch.float32, device=dev)                # no signals: one dummy phase
        d = _lib.NetDesc(R, t.n_lanes, t.n_cells, t.T, 1, max(int(t.T), 1), 1, float(dt), float(u_max), 0.2, float(vehicle_length))
        tc = t.c(0)
        lib = _lib.lib()
        ws_n = lib.dhts_net_hybrid_workspace_bytes(C.byref(d), C.byref(tc))
        if ws_n == 0:
            raise ValueError("unsupported hybrid network size")
        hist = torch.empty(max(R * (t.T + 1) * 4 * t.n_cells, 64), dtype=torch.float32, device=dev)
        tape = torch.empty(max(lib.dhts_net_hybrid_tape_bytes(C.byref(d)) // 4, 64), dtype=torch.float32, device=dev)
        kc = torch.empty(max(R * t.T * t.n_cells, 64), dtype=torch.float32, device=dev)
        queue = torch.empty(R, t.T, t.n_lanes, dtype=torch.float32, device=dev)
        reward = torch.empty(R, dtype=torch.float32, device=dev)
        counts = torch.zeros(R, 4, dtype=torch.int32, device=dev)
        veh = torch.zeros(R, 128, 4, dtype=torch.float32, device=dev)
        veh[:, :, 0] = -1.0
        events = torch.full((R, 256, 2), -1, dtype=torch.int32, device=dev)
        ws = torch.empty(ws_n, dtype=torch.uint8, device=dev)
        err = new_error_record(dev)
        io = _lib.HybridStateIO(1 if plain else 0, _ptr(state0), None if g0 is None else _ptr(g0), _ptr(veh), _ptr(events))
        check(lib.dhts_net_hybrid_state_rollout_fwd(C.byref(d), C.byref(tc), C.byref(io), _ptr(action), _ptr(hist), _ptr(tape), _ptr(kc),
                                                    _ptr(queue), _ptr(reward), _ptr(counts), _ptr(ws), _ptr(err), _stream()),
              "dhts_net_hybrid_state_rollout_fwd")
        if check_faults:
            raise_on_fault(err)
        fin = hist[:R * (t.T + 1) * 4 * t.n_cells].view(R, t.T + 1, 4, t.n_cells)[:, t.T]
        rT, yT, uT = fin[:, 0].clone(), fin[:, 1].clone(), fin[:, 2].clone()
        ctx.d, ctx.tables, ctx.plain, ctx.u_max, ctx.check_faults = d, t, bool(plain), float(u_max), bool(check_faults)
        ctx.save_for_backward(action, hist, tape, kc, queue, ws, r0c, u0c, rT, yT)
        ctx.mark_non_differentiable(yT, events, counts)
        return rT, yT, uT, veh, events, counts

    @staticmethod
    def backward(ctx, g_rT, _g_yT, g_uT, g_veh, _g_ev, _g_counts):
        action, hist, tape, kc, queue, ws, r0c, u0c, rT, yT = ctx.saved_tensors
        d, t = ctx.d, ctx.tables
        R, Cc = r0c.shape
        dev = r0c.device
        z = lambda: torch.zeros(R, Cc, dtype=torch.float32, device=dev)      # noqa: E731
        g_r = g_rT.contiguous().float() if g_rT is not None else z()
        g_u = g_uT.contiguous().float() if g_uT is not None else z()
        g_stateT = torch.stack([g_r, z(), g_u], dim=1).contiguous()          # the kernel turns the speed's cotangent into (r, y)'s itself
        gv = None
        if g_veh is not None:
            gv = g_veh[:, :, 1:3].contiguous().float()                        # cotangent of (position, speed)
        g_reward = torch.zeros(R, dtype=torch.float32, device=dev)            # a pure state tap
        g_action = torch.empty_like(action)
        g_state0 = torch.empty(R, 3, Cc, dtype=torch.float32, device=dev)
        err = new_error_record(dev)
        tc = t.c(0)
        check(_lib.lib().dhts_net_hybrid_state_rollout_bwd(C.byref(d), C.byref(tc), 1 if ctx.plain else 0, _ptr(action), _ptr(hist), _ptr(tape),
                                                           _ptr(kc), _ptr(queue), _ptr(g_reward), _ptr(g_stateT), _ptr(gv), _ptr(g_action),
                                                           _ptr(g_state0), _ptr(ws), _ptr(err), _stream()),
              "dhts_net_hybrid_state_rollout_bwd")
        if ctx.check_faults:
            raise_on_fault(err)
        # initial (r, y, u) -> (r0, u0): y0 = r0 (u0 - u_eq(r0)) (FullQ.from_r_u), u0 itself where step 0 read the given speed
        g_r0 = g_state0[:, 0].contiguous()
        g_y0 = g_state0[:, 1].contiguous()
        g_u0 = macro_state_from_ru_bwd(r0c, u0c, g_y0, g_r0, ctx.u_max)
        g_u0 = g_u0 + g_state0[:, 2]
        return g_r0, g_u0, None, None, None, None, None, None, None


def net_hybrid_state_rollout(r0, u0, dev_tables, dt, u_max, ghost0=None, plain=True, vehicle_length=5.0, check_faults=True):
    """(rT, yT, uT [R][C], veh [R][128][4] = (lane or -1, position, speed, a) per spawned vehicle, events [R][256][2] = (step, kind),
    counts [R][4] = (spawned, deposited, records, events)); differentiable in r0, u0 through rT, uT and veh[..., 1:3]."""
    return NetHybridStateRollout.apply(r0, u0, ghost0, dev_tables, float(dt), float(u_max), bool(plain), float(vehicle_length),
                                       bool(check_faults))


def net_hybrid_plan(n_replicas, n_action, dev_tables, n_inter_sq=1, frames_per_phase=1, dt=1.0 / 30.0, u_max=30.0):
    """What net_hybrid_rollout would launch for a batch of `n_replicas` under the current DHTS_OPT_HYB_PACK (dhts_net_hybrid_plan):
    {"packed": two replicas per compute unit, "block", "stage_h", "lds_fwd", "lds_bwd", "loc_lanes", "max_step_records", "cus"}."""
    t = dev_tables
    d = _lib.NetDesc(int(n_replicas), t.n_lanes, t.n_cells, t.T, int(n_inter_sq), int(frames_per_phase), int(n_action), float(dt),
                     float(u_max), 0.2, 5.0)
    tc = t.c(0)
    plan = (C.c_int32 * 8)()
    check(_lib.lib().dhts_net_hybrid_plan(C.byref(d), C.byref(tc), plan), "dhts_net_hybrid_plan")
    keys = ("packed", "block", "stage_h", "lds_fwd", "lds_bwd", "loc_lanes", "max_step_records", "cus")
    out = dict(zip(keys, [int(x) for x in plan]))
    out["packed"] = bool(out["packed"])
    return out


def net_hybrid_eval(action, dev_tables, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0, err=None):
    """An evaluation episode (hard thresholds, see net_macro_eval) of R replicas of a hybrid network: returns
    (reward [R], queue [R][T][L], counts [R][4] = vehicles spawned, vehicles deposited, 0, 0)."""
    a = _f32c(action.detach(), "action")
    R, A = a.shape
    t = dev_tables
    if t.n_replica_tables not in (0, R):
        raise ValueError("per-replica tables must match the number of replicas")
    d = _lib.NetDesc(R, t.n_lanes, t.n_cells, t.T, int(n_inter_sq), int(frames_per_phase), A, float(dt), float(u_max),
                     float(static_speed), float(vehicle_length))
    tc = t.c(0)
    queue = torch.empty(R, t.T, t.n_lanes, dtype=torch.float32, device=a.device)
    reward = torch.empty(R, dtype=torch.float32, device=a.device)
    counts = torch.zeros(R, 4, dtype=torch.int32, device=a.device)
    own_err = err is None
    if own_err:
        err = new_error_record(a.device)
    check(_lib.lib().dhts_net_hybrid_rollout_eval(C.byref(d), C.byref(tc), _ptr(a), _ptr(queue), _ptr(reward), _ptr(counts),
                                                  _ptr(err), _stream()), "dhts_net_hybrid_rollout_eval")
    if own_err:
        raise_on_fault(err)
    return reward, queue, counts


def net_hybrid_rollout(action, dev_tables, n_inter_sq, frames_per_phase, dt, u_max, static_speed=0.2, vehicle_length=5.0,
                       loss_steps=0, check_faults=True, err=None, err_bwd=None):
    """Returns (reward restricted to the first loss_steps steps [differentiable], full reward, queue [R][T][L], counts [R][4]).
    check_faults=False: nothing is read back in either direction (no host sync; usable inside a HIP-graph capture), and a
    non-finite cotangent in the reverse sweep (the reference asserts on it, dmacro_lane.py:308; it
    happens e.g. when a head gap clamps to exactly 0 and the IDM Jacobian divides by it, didm.py:60-70) is left in the
    returned gradient of that replica instead of raising, so that a batch survives one bad member.
    err: a caller-owned sticky fault record (new_error_record) used by both directions instead of a fresh one per call; the
    caller reads it when it wants to (raise_on_fault) -- a training loop checks once per so many iterations, not twice per pass.
    err_bwd: a second caller-owned record for the reverse sweep alone, for callers that tolerate its NaN fault (a batch with one
    bad member) but not the forward's CFL / capacity faults: the first fault wins a record, so the two must not share one."""
    return NetHybridRollout.apply(action, dev_tables, int(n_inter_sq), int(frames_per_phase), float(dt), float(u_max),
                                  float(static_speed), float(vehicle_length), int(loss_steps), bool(check_faults), err, err_bwd)
