// idm_device.hpp -- device math of the IDM car-following step (gfx950).
//
// Implements model/micro/_idm.py:6-50 (acceleration with the two clips) and model/micro/didm.py:13-103
// (2x2 Jacobians wrt the ego and the leading vehicle) for one vehicle, in double on float32 state, with the
// Jacobian entries rounded to float32 (the reference assembles them in th.zeros((2, 2))).
// Powers are the reference's integer ones: x**2, x**3, x**4 by multiplication, x**0.5 = sqrt.
#pragma once
#include <hip/hip_runtime.h>

#include "fast_math.hpp"

namespace dhts {

struct IdmParams {      // road/vehicle/micro_vehicle.py:21-28, kept as doubles like the reference's Python floats
    double a_max, a_pref, v_target, min_space, time_pref, length;
};

struct IdmStep {
    float np, nv;       // next position / speed, float32 store (_micro_lane.py:182-183, :280-285)
    float dE[4];        // d(p', v')/d(p, v) of the ego vehicle, row-major
    float dLd[4];       // d(p', v')/d(p, v) of the leading vehicle
    bool collided;      // raw gap < 0 (_micro_lane.py:188-192)
    // exported by the known-answer entry point only (dead code in the rollout kernel)
    double acc, sstar;
    bool clipped_acc, clipped_spacing;
};

// Reference-order version: IEEE double division / square root exactly where the reference divides and takes powers
// (validation; dhts_idm_batch variant 1).
// p, v: ego state; dp_raw, dv_raw: gap and speed difference to the leader as compute_state_delta returns them
// (_micro_lane.py:195-214).
__device__ __forceinline__ void idm_step_ieee(double p, double v, double dp_raw, double dv_raw, const IdmParams &m,
                                              double dt, IdmStep &o) {
    double dp = dp_raw, dv = dv_raw;
    o.collided = dp < 0;
    if (o.collided) { dp = 0; dv = 0; }                    // :151-160 "Set deltas to 0"
    const double dpc = (1e-5 > dp) ? 1e-5 : dp;            // :166 max(position_delta, POSITION_DELTA_EPS)

    // IDM.compute_acceleration, _idm.py:30-50
    const double two_sqrt_ab = 2 * sqrt(m.a_max * m.a_pref);
    double s = (m.min_space + v * m.time_pref + ((v * dv) / two_sqrt_ab));
    const bool clipped_s = (s < 0.0);
    s = (0. > s) ? 0. : s;
    const double vr = v / m.v_target;
    const double vr2 = vr * vr;
    const double sr = s / dpc;
    double acc = m.a_max * (1.0 - vr2 * vr2 - sr * sr);
    const double floor_acc = -v / dt;
    const bool clipped_a = (acc < floor_acc);
    acc = (floor_acc > acc) ? floor_acc : acc;

    o.np = (float)(p + dt * v);
    o.nv = (float)(v + dt * acc);
    o.acc = acc; o.sstar = s; o.clipped_acc = clipped_a; o.clipped_spacing = clipped_s;

    // dIDM.compute_dEgo / compute_dLeading, didm.py:38-103, with the UN-clamped deltas (dmicro_lane.py:97)
    o.dE[0] = 1.f; o.dE[1] = (float)dt; o.dE[2] = 0.f; o.dE[3] = 0.f;
    o.dLd[0] = o.dLd[1] = o.dLd[2] = o.dLd[3] = 0.f;
    if (!clipped_a) {
        const double dp2 = dp_raw * dp_raw;
        const double dp3 = dp2 * dp_raw;
        const double s2_dp3 = (s * s) / dp3;
        const double vt2 = m.v_target * m.v_target;
        const double free_term = -4.0 * ((v * v * v) / (vt2 * vt2));
        const double s_dp2 = s / dp2;
        o.dE[2] = (float)(dt * (-2 * m.a_max * s2_dp3));
        o.dLd[2] = (float)(dt * (2 * m.a_max * s2_dp3));
        if (clipped_s) {
            o.dE[3] = (float)(1 + dt * m.a_max * free_term);
            o.dLd[3] = (float)(dt * m.a_max * (-2 * s_dp2));
        } else {
            o.dE[3] = (float)(1 + dt * m.a_max * (free_term - 2 * s_dp2 * (m.time_pref + ((v + dv_raw) / two_sqrt_ab))));
            o.dLd[3] = (float)(dt * m.a_max * (-2 * s_dp2 * (-v / two_sqrt_ab)));
        }
    }
}

// itscp `micro` mode, differentiable episodes: the reference's lanes are plain MicroLane objects there (example/control/itscp/_env.py:484-498)
// whose vehicle states become float32 TENSORS with the first head gap, so IDM.compute_acceleration and the Euler step
// (_idm.py:30-50, _micro_lane.py:166-183) are evaluated by torch: every operation rounds to float32, a Python float operand is cast
// to float32 first, pow(x, 2.0) is x * x, pow(x, 4.0) is powf (here: the float64 power rounded once).  Values in that ladder;
// the Jacobian blocks -- what autograd differentiates: the forward's OWN operations -- from the analytic formulas at the operands the
// forward used: after a collision both deltas are the Python ints 0, 0 (_micro_lane.py:151-162) and below POSITION_DELTA_EPS max()
// picks the Python float (:166) -- constants, nothing flows through them (dIDM's un-clamped deltas, dmicro_lane.py:97, are the hybrid
// lanes' rule: there a zero gap divides by zero, here it must not -- reference runs of congested 8-second episodes have finite
// gradients, tests/golden/itscp_micro_jam_*.npz); abs() hands the gap's cotangent on with the sign of (leader - ego).
// dp_raw, dv_raw: gap and speed difference as float32 tensor arithmetic gives them (|p_l - p| - float32((len_l + len) / 2), v - v_l);
// lead_sign: sign(p_l - p) of a follower (1 for a head vehicle: its gap is handed in).
__device__ __forceinline__ void idm_step_f32(float p, float v, float dp_raw, float dv_raw, const IdmParams &m, double dt, IdmStep &o,
                                             float lead_sign = 1.f) {
    float dp = dp_raw, dv = dv_raw;
    o.collided = dp < 0.f;
    if (o.collided) { dp = 0.f; dv = 0.f; }
    const float dpc = ((float)1e-5 > dp) ? (float)1e-5 : dp;
    const double two_sqrt_ab = 2 * sqrt(m.a_max * m.a_pref);
    const float dtf = (float)dt;
    float s = ((float)m.min_space + v * (float)m.time_pref) + ((v * dv) / (float)two_sqrt_ab);
    const bool clipped_s = (s < 0.0f);
    s = clipped_s ? 0.0f : s;
    const float t1 = v / (float)m.v_target;
    const double t1d = (double)t1, t1d2 = t1d * t1d;
    const float p1 = (float)(t1d2 * t1d2);
    const float t2 = s / dpc;
    const float p2 = t2 * t2;
    float acc = (float)m.a_max * ((1.0f - p1) - p2);
    const float floor_acc = (-v) / dtf;
    const bool clipped_a = (acc < floor_acc);
    acc = (floor_acc > acc) ? floor_acc : acc;
    o.np = p + dtf * v;
    o.nv = v + dtf * acc;
    o.acc = acc; o.sstar = s; o.clipped_acc = clipped_a; o.clipped_spacing = clipped_s;
    o.dE[0] = 1.f; o.dE[1] = (float)dt; o.dE[2] = 0.f; o.dE[3] = 0.f;
    o.dLd[0] = o.dLd[1] = o.dLd[2] = o.dLd[3] = 0.f;
    if (!clipped_a) {
        const bool live_dp = !o.collided && !((float)1e-5 > dp);      // the gap is still the tensor
        const double vd = v, sd = s, dpr = dpc, dvr = o.collided ? -vd : (double)dv_raw;      // (collided: v + dv = 0 below)
        const double dp2 = dpr * dpr;
        const double dp3 = dp2 * dpr;
        const double s2_dp3 = (sd * sd) / dp3;
        const double vt2 = m.v_target * m.v_target;
        const double free_term = -4.0 * ((vd * vd * vd) / (vt2 * vt2));
        const double s_dp2 = sd / dp2;
        o.dE[2] = live_dp ? (float)(dt * (-2 * m.a_max * s2_dp3)) * lead_sign : 0.f;
        o.dLd[2] = live_dp ? (float)(dt * (2 * m.a_max * s2_dp3)) * lead_sign : 0.f;
        if (clipped_s) {
            o.dE[3] = (float)(1 + dt * m.a_max * free_term);
            o.dLd[3] = (float)(dt * m.a_max * (-2 * s_dp2));
        } else {
            o.dE[3] = (float)(1 + dt * m.a_max * (free_term - 2 * s_dp2 * (m.time_pref + ((vd + dvr) / two_sqrt_ab))));
            o.dLd[3] = (float)(dt * m.a_max * (-2 * s_dp2 * (-vd / two_sqrt_ab)));
        }
        if (o.collided) o.dLd[3] = 0.f;        // (speed_delta is the int 0: the leader's speed is out of the step)
    }
}

// A lane of a hybrid network: followers and head vehicle behind ONE instruction stream (they share wavefronts: as two functions every
// such wavefront ran both bodies).
// mixed = false: idm_step_ieee's arithmetic, bit for bit (followers; every vehicle of an evaluation episode).
// mixed = true: the HEAD vehicle of a dMicroLane in a differentiable itscp hybrid episode.  dMicroLane.detach_vehicle (dmicro_lane.py:228-250)
// turns the vehicles' states into Python floats but leaves the lane's head gap alone, and there the gap is a float32 TENSOR (the signal
// blend of example/control/itscp/_simulator.py:260-263) -- so for this one vehicle IDM.compute_acceleration and the Euler step run in MIXED
// arithmetic: what combines two Python floats is double, what meets the tensor is float32 (the Python operand cast first),
// pow(tensor, 2.0) is x * x; the position update stays double.  p, v: the vehicle's float32 values (Python floats there); dp_raw, dv_raw:
// the gap and speed difference (the tensor's float32 values, widened).
// The square root, the free-road term and the Jacobian blocks (analytic, at the same operands) are common; the acceleration and the new
// speed are evaluated both ways -- the float32 side is a dozen cheap instructions -- and `mixed` selects.
__device__ __forceinline__ void idm_step_lane(float p, float v, double dp_raw, double dv_raw, bool mixed, const IdmParams &m, double dt, IdmStep &o) {
    double dp = dp_raw, dv = dv_raw;
    o.collided = dp < 0;
    if (o.collided) { dp = 0; dv = 0; }
    const double dpc = (1e-5 > dp) ? 1e-5 : dp;
    const double vd = v;
    const double two_sqrt_ab = 2 * sqrt(m.a_max * m.a_pref);
    const double vr = vd / m.v_target, vr2 = vr * vr;
    const double floor_acc = -vd / dt;
    // double (followers; every vehicle of an evaluation episode)
    double s_d = (m.min_space + vd * m.time_pref + ((vd * dv) / two_sqrt_ab));
    const bool clipped_s_d = (s_d < 0.0);
    s_d = (0. > s_d) ? 0. : s_d;
    const double sr = s_d / dpc;
    double acc_d = m.a_max * (1.0 - vr2 * vr2 - sr * sr);
    const bool clipped_a_d = (acc_d < floor_acc);
    acc_d = (floor_acc > acc_d) ? floor_acc : acc_d;
    const float nv_d = (float)(vd + dt * acc_d);
    // mixed (the head vehicle under a float32 tensor gap)
    const float dpf = (float)dp, dvf = (float)dv;
    const float dpcf = ((float)1e-5 > dpf) ? (float)1e-5 : dpf;
    float s_f = (float)(m.min_space + vd * m.time_pref) + ((v * dvf) / (float)two_sqrt_ab);
    const bool clipped_s_f = (s_f < 0.0f);
    s_f = clipped_s_f ? 0.0f : s_f;
    const float t2 = s_f / dpcf;
    const float acc_f = (float)m.a_max * ((float)(1.0 - vr2 * vr2) - (t2 * t2));
    const bool clipped_a_f = (acc_f < (float)floor_acc);
    const float nv_f = ((float)floor_acc > acc_f) ? (float)(vd + dt * floor_acc) : v + ((float)dt * acc_f);

    const double s = mixed ? (double)s_f : s_d;
    const bool clipped_s = mixed ? clipped_s_f : clipped_s_d, clipped_a = mixed ? clipped_a_f : clipped_a_d;
    o.np = (float)((double)p + dt * vd);
    o.nv = mixed ? nv_f : nv_d;
    o.acc = mixed ? (double)acc_f : acc_d; o.sstar = s; o.clipped_acc = clipped_a; o.clipped_spacing = clipped_s;
    o.dE[0] = 1.f; o.dE[1] = (float)dt; o.dE[2] = 0.f; o.dE[3] = 0.f;
    o.dLd[0] = o.dLd[1] = o.dLd[2] = o.dLd[3] = 0.f;
    if (!clipped_a) {
        const double dp2 = dp_raw * dp_raw;
        const double dp3 = dp2 * dp_raw;
        const double s2_dp3 = (s * s) / dp3;
        const double vt2 = m.v_target * m.v_target;
        const double free_term = -4.0 * ((vd * vd * vd) / (vt2 * vt2));
        const double s_dp2 = s / dp2;
        o.dE[2] = (float)(dt * (-2 * m.a_max * s2_dp3));
        o.dLd[2] = (float)(dt * (2 * m.a_max * s2_dp3));
        if (clipped_s) {
            o.dE[3] = (float)(1 + dt * m.a_max * free_term);
            o.dLd[3] = (float)(dt * m.a_max * (-2 * s_dp2));
        } else {
            o.dE[3] = (float)(1 + dt * m.a_max * (free_term - 2 * s_dp2 * (m.time_pref + ((vd + dv_raw) / two_sqrt_ab))));
            o.dLd[3] = (float)(dt * m.a_max * (-2 * s_dp2 * (-vd / two_sqrt_ab)));
        }
    }
}

// ---- production version -------------------------------------------------------------------------------------
// Per-vehicle constants evaluated once per rollout (IEEE operations), so that the time loop has no division by a
// constant and no square root: 1 / (2 sqrt(a_max a_pref)), 1 / v_target, 1 / v_target^4, and the products of the
// step size with a_max that the Jacobian entries start with.
struct IdmDerived {
    double a_max, min_space, time_pref, length;
    double inv_2sab, inv_vt, inv_vt4;
    double dt_a, m2_dt_a;           // dt a_max, -2 dt a_max   (set by idm_set_dt)
};
__device__ __forceinline__ IdmDerived idm_derive(const IdmParams &m) {
    IdmDerived d;
    d.a_max = m.a_max; d.min_space = m.min_space; d.time_pref = m.time_pref; d.length = m.length;
    d.inv_2sab = 1.0 / (2 * sqrt(m.a_max * m.a_pref));
    d.inv_vt = 1.0 / m.v_target;
    const double vt2 = m.v_target * m.v_target;
    d.inv_vt4 = 1.0 / (vt2 * vt2);
    d.dt_a = 0.; d.m2_dt_a = 0.;
    return d;
}
__device__ __forceinline__ void idm_set_dt(IdmDerived &d, double dt) { d.dt_a = dt * d.a_max; d.m2_dt_a = -2.0 * d.dt_a; }

// Same formulas and clip logic as idm_step_ieee; the only division left is ONE reciprocal of the gap.  Results differ
// from the reference-order version by a few double ulps (< 1e-8 of a float32 ulp before the float32 stores).
// Every vector instruction costs the same ~4 cycles on this chip whatever its type, so the step is written for
// instruction COUNT: constant factors folded per vehicle, the rare cases (collision, gap below 1e-5) behind wave-uniform
// branches instead of per-lane selects, and no separate formulas under the spacing clip -- with s* clipped to 0 the
// general Jacobian entries reduce to the clipped ones exactly (every s* term is a product with 0).  The three max() are
// v_max_f64 (a NaN operand yields the other one, where Python's max(x, c) keeps a NaN x): a NaN state stays NaN through
// p' = p + dt v and v' = v + dt acc all the same.
__device__ __forceinline__ void idm_step(double p, double v, double dp_raw, double dv_raw, const IdmDerived &m,
                                         double dt, double inv_dt, IdmStep &o) {
    double dp = dp_raw, dv = dv_raw;
    o.collided = dp < 0;
    if (__builtin_amdgcn_ballot_w64(o.collided)) {          // :151-160 "Set deltas to 0" -- rare: wave-uniform branch
        asm volatile("" ::: "memory");                      // (keeps it a branch: no if-conversion into the hot path)
        if (o.collided) { dp = 0; dv = 0; }
    }
    const double dpc = fmax(dp, 1e-5);                      // :166 max(position_delta, POSITION_DELTA_EPS)
    const double rdp = fast_rcp(dpc);

    double s = __builtin_fma(v * dv, m.inv_2sab, __builtin_fma(v, m.time_pref, m.min_space));
    const bool clipped_s = (s < 0.0);
    s = fmax(s, 0.);
    const double vr = v * m.inv_vt;
    const double vr2 = vr * vr;
    const double sr = s * rdp;
    double acc = m.a_max * __builtin_fma(-sr, sr, __builtin_fma(-vr2, vr2, 1.0));
    const double floor_acc = -v * inv_dt;
    const bool clipped_a = (acc < floor_acc);
    acc = fmax(acc, floor_acc);

    o.np = (float)__builtin_fma(dt, v, p);
    o.nv = (float)(v + dt * acc);          // not fused: under the acceleration clip dt * (-v / dt) rounds back to -v and the vehicle stops at exactly 0
    o.acc = acc; o.sstar = s; o.clipped_acc = clipped_a; o.clipped_spacing = clipped_s;

    // the Jacobians use the UN-clamped gap (dmicro_lane.py:97); it equals the clamped one unless gap < 1e-5
    double rdr = rdp;
    if (__builtin_amdgcn_ballot_w64(!(dp_raw >= 1e-5))) {   // rare: wave-uniform branch around the IEEE division
        asm volatile("" ::: "memory");                      // (without it the compiler flattens the branch: 14 instructions of
        if (!(dp_raw >= 1e-5)) rdr = 1.0 / dp_raw;          //  v_div_scale / fmas / fixup on every vehicle-step)
    }
    const double rdr2 = rdr * rdr;
    const double s_dp2 = s * rdr2;
    const double e2 = m.m2_dt_a * ((s_dp2 * s) * rdr);                    // dt (-2 a s^2 / gap^3)
    const double free4 = -4.0 * ((v * v * v) * m.inv_vt4);
    const double tl = __builtin_fma(v + dv_raw, m.inv_2sab, m.time_pref); // T + (v + dv) / (2 sqrt(ab))
    const double e3 = __builtin_fma(m.dt_a, __builtin_fma(-2.0 * s_dp2, tl, free4), 1.0);
    const double l3 = (m.m2_dt_a * s_dp2) * (-v * m.inv_2sab);
    o.dE[0] = 1.f; o.dE[1] = (float)dt;
    o.dE[2] = clipped_a ? 0.f : (float)e2;
    o.dE[3] = clipped_a ? 0.f : (float)e3;
    o.dLd[0] = o.dLd[1] = 0.f;
    o.dLd[2] = -o.dE[2];
    o.dLd[3] = clipped_a ? 0.f : (float)l3;
}

}  // namespace dhts
