// micro_kernels.hip -- time-fused forward and reverse sweeps of the IDM car-following ODE on gfx950.
//
// Forward (micro_rollout_fwd_kernel<K>): one 64-lane wavefront owns one traffic lane for the whole rollout.
//   Vehicle slot i follows slot i + 1 (MicroLane.curr_vehicle order), so the leader of a vehicle is simply the
//   next slot: positions and speeds (float32) live in LDS, thread t of pass j owns slot 64 j + t, its six
//   double parameters stay in registers for all T steps (K = passes is a template parameter so they do), and
//   passes run tail -> head so that a pass reads its leaders' OLD state before the next pass overwrites it.
//   HBM sees the initial load, the final store and the Jacobian tape (32 B per vehicle-step, two coalesced
//   16-B-per-lane streams).  The head vehicle uses the lane's (head_position_delta, head_speed_delta).
// Reverse (micro_rollout_bwd_kernel): one workgroup per lane, cotangent in LDS, newest step first:
//   g'[i] = dEgo[i]^T g[i] + dLeading[i-1]^T g[i-1]; the virtual-leader slot's cotangent returns to the head
//   vehicle and to the head gap (dmicro_lane.py:130-153, 271-298).
//
// Reference: road/lane/_micro_lane.py:131-214, road/lane/dmicro_lane.py:87-127 and :271-298.
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"
#include "arz_device.hpp"   // dot2
#include "idm_device.hpp"

namespace dhts {

__device__ __forceinline__ void raise_fault_m(dhts_error *err, int code, int step, int lane, int index) {
    if (err == nullptr) return;
    if (atomicCAS(&err->code, 0, code) == 0) {
        err->step = step;
        err->lane = lane;
        err->index = index;
    }
}

// compact rollout tape entry: (dEgo[1][0], dEgo[1][1], dLeading[1][1]).  The first rows of both blocks are constants
// ([1, dt] and [0, 0]) and dLeading[1][0] = -dEgo[1][0] bit for bit: dt * (2 a s^2 / gap^3) against dt * (-2 a s^2 / gap^3),
// both 0 under the acceleration clip (didm.py:38-103) -- 12 bytes per vehicle-step instead of the 32 of dqs[V][2][2][2]
struct __attribute__((packed, aligned(4))) MicroTape3 { float e2, e3, l3; };

// grid = L workgroups of 64 * kW threads; dynamic LDS = 2 * (V + 1) floats (kW = 1) or twice that (kW > 1: the state
// ping-pongs between two buffers so that one workgroup barrier per step separates a step's reads from the next step's).
// kW wavefronts per lane: thread `tid` of pass j owns slot (64 kW) j + tid.
// kFull: every slot of every lane holds a vehicle (no counts, V = 64 kW K): no validity masks, no index clamps, and only the
// last pass can hold the head vehicle.
template <int K, int kW, bool kCompact, bool kFull>
__global__ __launch_bounds__(64 * kW) void micro_rollout_fwd_kernel(
    int L, int V, int T, double dt,
    const float *__restrict__ p_in, const float *__restrict__ v_in, const int32_t *__restrict__ count,
    const double *__restrict__ params, const double *__restrict__ head,
    float *__restrict__ p_out, float *__restrict__ v_out, float *__restrict__ tape, float *__restrict__ hist,
    dhts_error *err) {
    extern __shared__ float lds[];
    const int lane = blockIdx.x;
    const int t = threadIdx.x;
    constexpr int kStride = 64 * kW;
    float *Sp = lds, *Sv = lds + (V + 1);
    const size_t base = (size_t)lane * V;
    const int n = (!kFull && count) ? count[lane] : V;
    const int Vp = (V + 63) & ~63;
    const size_t plane = (size_t)L * V;

    IdmDerived prm[K];
    double len_lead[K];
    const double inv_dt = 1.0 / dt;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int i = j * kStride + t;
        const int ic = i < V ? i : V - 1;
        const int il = (i + 1) < V ? (i + 1) : V - 1;
        IdmParams raw;
        raw.a_max = params[0 * plane + base + ic];
        raw.a_pref = params[1 * plane + base + ic];
        raw.v_target = params[2 * plane + base + ic];
        raw.min_space = params[3 * plane + base + ic];
        raw.time_pref = params[4 * plane + base + ic];
        raw.length = params[5 * plane + base + ic];
        prm[j] = idm_derive(raw);
        idm_set_dt(prm[j], dt);
        len_lead[j] = params[5 * plane + base + il];
    }
    for (int k = t; k < V; k += kStride) { Sp[k] = p_in[base + k]; Sv[k] = v_in[base + k]; }
    if (t == 0) { Sp[V] = 0.f; Sv[V] = 0.f; if (kW > 1) { Sp[2 * (V + 1) + V] = 0.f; Sv[2 * (V + 1) + V] = 0.f; } }
    __syncthreads();
    const double head_dp = head[(size_t)lane * 2], head_dv = head[(size_t)lane * 2 + 1];
    int fault_step = -1, fault_index = 0;

    for (int step = 0; step < T; ++step) {
        float4 *tp = (tape && !kCompact) ? reinterpret_cast<float4 *>(tape) + ((size_t)step * L + lane) * 2 * Vp : nullptr;
        MicroTape3 *tc = (tape && kCompact) ? reinterpret_cast<MicroTape3 *>(tape) + ((size_t)step * L + lane) * Vp : nullptr;
        float *hp = hist ? hist + ((size_t)step * L + lane) * 2 * V : nullptr;
        // every pass reads its own and its leaders' OLD state before anything is written, so the K IDM evaluations of a
        // thread are independent instruction streams (a wave's LDS operations execute in order: no barrier is needed
        // between this step's writes and the next step's reads)
        // kW > 1: read buffer (step & 1), write the other one
        const float *Rp = Sp + ((kW > 1 && (step & 1)) ? 2 * (V + 1) : 0), *Rv = Sv + ((kW > 1 && (step & 1)) ? 2 * (V + 1) : 0);
        float *Wp = Sp + ((kW > 1 && !(step & 1)) ? 2 * (V + 1) : 0), *Wv = Sv + ((kW > 1 && !(step & 1)) ? 2 * (V + 1) : 0);
        float rp[K], rv[K], rpl[K], rvl[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = j * kStride + t;
            const int ic = (kFull || i < n) ? i : 0;
            rp[j] = Rp[ic]; rv[j] = Rv[ic]; rpl[j] = Rp[ic + 1]; rvl[j] = Rv[ic + 1];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = j * kStride + t;
            const bool valid = kFull || i < n;
            const double p = rp[j], v = rv[j];
            const double pl = rpl[j], vl = rvl[j];
            double dp, dv;
            if ((!kFull || j == K - 1) && i == n - 1) {      // compute_state_delta, _micro_lane.py:201-204
                dp = head_dp; dv = head_dv;
            } else {                                  // :206-212
                dp = fabs(pl - p) - ((len_lead[j] + prm[j].length) * 0.5);
                dv = v - vl;
            }
            IdmStep o;
            idm_step(p, v, dp, dv, prm[j], dt, inv_dt, o);
            if (valid) {
                if (o.collided && fault_step < 0) { fault_step = step; fault_index = i; }
                Wp[i] = o.np; Wv[i] = o.nv;
                if constexpr (kCompact) {
                    if (tc) { MicroTape3 e; e.e2 = o.dE[2]; e.e3 = o.dE[3]; e.l3 = o.dLd[3]; tc[i] = e; }
                } else if (tp) {
                    tp[i] = make_float4(o.dE[0], o.dE[1], o.dE[2], o.dE[3]);
                    tp[Vp + i] = make_float4(o.dLd[0], o.dLd[1], o.dLd[2], o.dLd[3]);
                }
                if (hp) { hp[i] = o.np; hp[V + i] = o.nv; }
            }
        }
        if constexpr (kW > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS-only: tape stores stay in flight
    }
    __syncthreads();
    const float *Fp = Sp + ((kW > 1 && (T & 1)) ? 2 * (V + 1) : 0), *Fv = Sv + ((kW > 1 && (T & 1)) ? 2 * (V + 1) : 0);
    // slots >= count pass through: a step writes only live slots, and with kW > 1 and an odd T the final buffer is the one the
    // initial load never filled
    for (int k = t; k < V; k += kStride) { p_out[base + k] = k < n ? Fp[k] : p_in[base + k]; v_out[base + k] = k < n ? Fv[k] : v_in[base + k]; }
    if (fault_step >= 0) raise_fault_m(err, DHTS_FAULT_COLLISION, fault_step, lane, fault_index);
}

// known-answer entry: n independent vehicles.  in [9][n] double = a_max a_pref v v_target dp dv min_space time_pref dt
__global__ void idm_batch_kernel(int64_t n, int variant, const double *__restrict__ in, double *__restrict__ next_pv,
                                 float *__restrict__ dE, float *__restrict__ dLd, int32_t *__restrict__ collided,
                                 double *__restrict__ acc_s, int32_t *__restrict__ clips) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        IdmParams m;
        m.a_max = in[i]; m.a_pref = in[n + i]; m.v_target = in[3 * n + i]; m.min_space = in[6 * n + i];
        m.time_pref = in[7 * n + i]; m.length = 0.;
        IdmStep o;
        const double dt = in[8 * n + i];
        if (variant == 1) idm_step_ieee(0.0, in[2 * n + i], in[4 * n + i], in[5 * n + i], m, dt, o);
        else { IdmDerived dm = idm_derive(m); idm_set_dt(dm, dt); idm_step(0.0, in[2 * n + i], in[4 * n + i], in[5 * n + i], dm, dt, 1.0 / dt, o); }
        next_pv[i] = o.np; next_pv[n + i] = o.nv;
        collided[i] = o.collided ? 1 : 0;
        acc_s[i] = o.acc; acc_s[n + i] = o.sstar;
        clips[i] = o.clipped_acc ? 1 : 0; clips[n + i] = o.clipped_spacing ? 1 : 0;
        for (int j = 0; j < 4; ++j) { dE[j * n + i] = o.dE[j]; dLd[j * n + i] = o.dLd[j]; }
    }
}

// known-answer entry of the Jacobians ALONE, with the caller's optimal spacing and clip flags (dIDM.compute_dEgo / compute_dLeading
// take them as arguments, didm.py:13-103: the reference's lane passes flags derived from the CLAMPED gap beside the un-clamped gap
// itself, dmicro_lane.py:97).  in [12][n] double = a_max a_pref v v_target dp dv min_space time_pref optimal_spacing dt
// clipped_acceleration clipped_optimal_spacing
__global__ void idm_jac_batch_kernel(int64_t n, const double *__restrict__ in, float *__restrict__ dE, float *__restrict__ dLd) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double a_max = in[i], a_pref = in[n + i], v = in[2 * n + i], v_t = in[3 * n + i], dp = in[4 * n + i], dv = in[5 * n + i];
        const double time_pref = in[7 * n + i], s = in[8 * n + i], dt = in[9 * n + i];
        const bool clipped_a = in[10 * n + i] != 0., clipped_s = in[11 * n + i] != 0.;
        float e[4] = {1.f, (float)dt, 0.f, 0.f}, l[4] = {0.f, 0.f, 0.f, 0.f};
        if (!clipped_a) {
            const double dp2 = dp * dp, dp3 = dp2 * dp;
            const double s2_dp3 = (s * s) / dp3, s_dp2 = s / dp2;
            const double vt2 = v_t * v_t;
            const double free_term = -4.0 * ((v * v * v) / (vt2 * vt2));
            const double two_sqrt_ab = 2 * sqrt(a_max * a_pref);
            e[2] = (float)(dt * (-2 * a_max * s2_dp3));
            l[2] = (float)(dt * (2 * a_max * s2_dp3));
            if (clipped_s) {
                e[3] = (float)(1 + dt * a_max * free_term);
                l[3] = (float)(dt * a_max * (-2 * s_dp2));
            } else {
                e[3] = (float)(1 + dt * a_max * (free_term - 2 * s_dp2 * (time_pref + ((v + dv) / two_sqrt_ab))));
                l[3] = (float)(dt * a_max * (-2 * s_dp2 * (-v / two_sqrt_ab)));
            }
        }
        for (int j = 0; j < 4; ++j) { dE[j * n + i] = e[j]; dLd[j * n + i] = l[j]; }
    }
}

// grid = L workgroups of blockDim.x threads; dynamic LDS = 4 * (V + 2) floats
template <bool kCompact>
__global__ void micro_rollout_bwd_kernel(
    int L, int V, int T, double dt, const float *__restrict__ tape, const int32_t *__restrict__ count,
    const float *__restrict__ g_p_in, const float *__restrict__ g_v_in, const float *__restrict__ g_hist,
    float *__restrict__ g_p_out, float *__restrict__ g_v_out, double *__restrict__ g_head, int fold, dhts_error *err) {
    extern __shared__ float lds[];
    const int lane = blockIdx.x;
    const int t = threadIdx.x;
    const int B = blockDim.x;
    const int P = V + 2;
    float *Gp = lds, *Gv = lds + P, *C1p = lds + 2 * P, *C1v = lds + 3 * P;   // C1[i + 1] = dLeading[i]^T g[i]
    const size_t base = (size_t)lane * V;
    const int n = count ? count[lane] : V;
    const int Vp = (V + 63) & ~63;

    for (int k = t; k < P; k += B) { Gp[k] = 0.f; Gv[k] = 0.f; C1p[k] = 0.f; C1v[k] = 0.f; }
    __syncthreads();
    for (int k = t; k < n; k += B) { Gp[k] = g_p_in[base + k]; Gv[k] = g_v_in[base + k]; }
    __syncthreads();

    double gh_p = 0., gh_v = 0.;       // held by the thread that owns the head vehicle
    int bad_step = -1, bad_index = 0;  // first non-finite cotangent this thread meets (the latest step: the sweep runs backwards)
    int step_hi = T - 1;
    if constexpr (kCompact) {
        if (V <= B && T > 0) {         // (T = 0: no tape to prefetch from -- the tape pointer may be NULL)
            // One vehicle per thread (the rollouts' common shape): the tape entry of the NEXT step to replay is loaded while
            // this step computes (unconditional load, clamped step index), and the barriers wait for LDS only, so the load
            // stays in flight across them -- the tape stream is what bounds this kernel.  Per-step cotangents (a loss on the
            // state history) travel the same way.
            const int k = t;
            const bool vk = k < n;
            const MicroTape3 *tc0 = reinterpret_cast<const MicroTape3 *>(tape) + (size_t)lane * Vp + (k < V ? k : 0);
            const size_t step_stride = (size_t)L * Vp;
            MicroTape3 nx = tc0[(size_t)(T - 1) * step_stride];
            MicroTape3 nx2 = tc0[(size_t)(T > 1 ? T - 2 : 0) * step_stride];
            const float *gh0 = g_hist ? g_hist + (size_t)lane * 2 * V + (k < V ? k : 0) : nullptr;
            const size_t h_stride = (size_t)L * 2 * V;
            float hp1 = 0.f, hv1 = 0.f, hp2 = 0.f, hv2 = 0.f;
            if (gh0) {
                const float *a = gh0 + (size_t)(T - 1) * h_stride, *b = gh0 + (size_t)(T > 1 ? T - 2 : 0) * h_stride;
                hp1 = a[0]; hv1 = a[V]; hp2 = b[0]; hv2 = b[V];
            }
            const float dtf = (float)dt;
            for (int step = T - 1; step >= 0; --step) {
                const MicroTape3 c = nx;
                const float chp = hp1, chv = hv1;
                nx = nx2; hp1 = hp2; hv1 = hv2;
                nx2 = tc0[(size_t)(step > 1 ? step - 2 : 0) * step_stride];       // two steps ahead
                if (gh0) { const float *a = gh0 + (size_t)(step > 1 ? step - 2 : 0) * h_stride; hp2 = a[0]; hv2 = a[V]; }
                if (vk) {
                    float gp = Gp[k], gv = Gv[k];
                    if (gh0) { gp += chp; gv += chv; }
                    Gp[k] = dot2(1.f, gp, c.e2, gv);           // grad_ps[:-1] = dqs[:, 0]^T g
                    Gv[k] = dot2(dtf, gp, c.e3, gv);
                    C1p[k + 1] = dot2(0.f, gp, -c.e2, gv);     // grad_ps[1:] += dqs[:, 1]^T g
                    C1v[k + 1] = dot2(0.f, gp, c.l3, gv);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (vk) {
                    float np_ = Gp[k], nv_ = Gv[k];
                    if (k > 0) { np_ += C1p[k]; nv_ += C1v[k]; }
                    if (k == n - 1) {
                        const float vp = C1p[n], vv = C1v[n];
                        if (fold) {
                            gh_p += (double)vp;
                            gh_v -= (double)vv;
                            np_ += vp; nv_ += vv;
                        } else {
                            gh_p = (double)vp;
                            gh_v = (double)vv;
                        }
                    }
                    Gp[k] = np_; Gv[k] = nv_;
                    // a gap clamped to 0 makes the IDM Jacobian divide by it (didm.py:60-70) and the sweep meet 0 * inf.  The
                    // reference's micro backward hands such NaNs on silently (dmicro_lane.py:271-298 has no assert, unlike
                    // dmacro_lane.py:308); the values are kept, and the fault record says where the first one appeared
                    if (bad_step < 0 && !(isfinite(np_) && isfinite(nv_))) { bad_step = step; bad_index = k; }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            step_hi = -1;
            __syncthreads();
        }
    }
    for (int step = step_hi; step >= 0; --step) {
        const float4 *tp = reinterpret_cast<const float4 *>(tape) + ((size_t)step * L + lane) * 2 * Vp;            // reference layout
        const MicroTape3 *tc = reinterpret_cast<const MicroTape3 *>(tape) + ((size_t)step * L + lane) * Vp;         // compact layout
        const float *gh = g_hist ? g_hist + ((size_t)step * L + lane) * 2 * V : nullptr;
        for (int k = t; k < n; k += B) {
            float4 dE, dLd;
            if constexpr (kCompact) {
                const MicroTape3 c = tc[k];
                dE = make_float4(1.f, (float)dt, c.e2, c.e3); dLd = make_float4(0.f, 0.f, -c.e2, c.l3);
            } else { dE = tp[k]; dLd = tp[Vp + k]; }
            float gp = Gp[k], gv = Gv[k];
            if (gh) { gp += gh[k]; gv += gh[V + k]; }
            Gp[k] = dot2(dE.x, gp, dE.z, gv);          // grad_ps[:-1] = dqs[:, 0]^T g
            Gv[k] = dot2(dE.y, gp, dE.w, gv);
            C1p[k + 1] = dot2(dLd.x, gp, dLd.z, gv);   // grad_ps[1:] += dqs[:, 1]^T g
            C1v[k + 1] = dot2(dLd.y, gp, dLd.w, gv);
        }
        __syncthreads();
        for (int k = t; k < n; k += B) {
            float np_ = Gp[k], nv_ = Gv[k];
            if (k > 0) { np_ += C1p[k]; nv_ += C1v[k]; }
            if (k == n - 1) {
                // virtual leader = (p_head + head_dp, v_head - head_dv): its cotangent C1[n] returns to the head
                const float vp = C1p[n], vv = C1v[n];
                if (fold) {
                    gh_p += (double)vp;
                    gh_v -= (double)vv;
                    np_ += vp; nv_ += vv;
                } else {                   // single-step operator form: hand the raw slot cotangent back
                    gh_p = (double)vp;
                    gh_v = (double)vv;
                }
            }
            Gp[k] = np_; Gv[k] = nv_;
            if (bad_step < 0 && !(isfinite(np_) && isfinite(nv_))) { bad_step = step; bad_index = k; }
        }
        __syncthreads();
    }
    for (int k = t; k < V; k += B) {
        g_p_out[base + k] = (k < n) ? Gp[k] : 0.f;
        g_v_out[base + k] = (k < n) ? Gv[k] : 0.f;
    }
    if (g_head) {
        if (n == 0) { if (t == 0) { g_head[(size_t)lane * 2] = 0.; g_head[(size_t)lane * 2 + 1] = 0.; } }
        else if (t == (n - 1) % B) { g_head[(size_t)lane * 2] = gh_p; g_head[(size_t)lane * 2 + 1] = gh_v; }
    }
    // the lane's report: the latest step at which a non-finite cotangent appeared (the first the sweep met), lowest vehicle
    __shared__ int s_bad;
    if (t == 0) s_bad = -1;
    __syncthreads();
    if (bad_step >= 0) atomicMax(&s_bad, (bad_step << 10) | (1023 - bad_index));      // capacity <= 1024 vehicles
    __syncthreads();
    if (t == 0 && s_bad >= 0) raise_fault_m(err, DHTS_FAULT_NAN, s_bad >> 10, lane, 1023 - (s_bad & 1023));
}

// The single-step operator in the float32 TENSOR ladder (idm_step_f32): what the reference's plain MicroLane computes when its vehicle
// states are torch tensors -- itscp `micro` mode, differentiable episodes (example/control/itscp/_env.py:484-498; _micro_lane.py:131-214
// evaluated by torch).  One thread per vehicle slot; same tape as dhts_micro_step_fwd ([lane][2][Vp][4]), same reverse operator.
// kHeadOnly: the head vehicle alone meets a float32 tensor (its gap) -- mixed arithmetic for it, double for the followers (idm_step_lane)
template <bool kHeadOnly>
__global__ void micro_step_tensor_fwd_kernel(int L, int V, double dt, const float *__restrict__ p_in, const float *__restrict__ v_in,
                                             const int32_t *__restrict__ count, const double *__restrict__ params,
                                             const double *__restrict__ head, float *__restrict__ p_out, float *__restrict__ v_out,
                                             float *__restrict__ tape, dhts_error *err) {
    const int lane = blockIdx.x;
    const size_t base = (size_t)lane * V, plane = (size_t)L * V;
    const int n = count ? count[lane] : V;
    const int Vp = (V + 63) & ~63;
    float4 *tp = tape ? reinterpret_cast<float4 *>(tape) + (size_t)lane * 2 * Vp : nullptr;
    int fault_index = -1;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        if (i >= n) { p_out[base + i] = p_in[base + i]; v_out[base + i] = v_in[base + i]; continue; }
        IdmParams m;
        m.a_max = params[0 * plane + base + i]; m.a_pref = params[1 * plane + base + i]; m.v_target = params[2 * plane + base + i];
        m.min_space = params[3 * plane + base + i]; m.time_pref = params[4 * plane + base + i]; m.length = params[5 * plane + base + i];
        const float pi = p_in[base + i], vi = v_in[base + i];
        float dp, dv;
        if (i == n - 1) { dp = (float)head[(size_t)lane * 2]; dv = (float)head[(size_t)lane * 2 + 1]; }
        else {
            const double len_l = params[5 * plane + base + i + 1];
            dp = fabsf(p_in[base + i + 1] - pi) - (float)((len_l + m.length) * 0.5);
            dv = vi - v_in[base + i + 1];
        }
        IdmStep o;
        if constexpr (kHeadOnly) {
            double dpd, dvd;
            if (i == n - 1) { dpd = (double)(float)head[(size_t)lane * 2]; dvd = (double)(float)head[(size_t)lane * 2 + 1]; }
            else {
                const double len_l = params[5 * plane + base + i + 1];
                dpd = fabs((double)p_in[base + i + 1] - (double)pi) - ((len_l + m.length) * 0.5);
                dvd = (double)vi - (double)v_in[base + i + 1];
            }
            idm_step_lane(pi, vi, dpd, dvd, i == n - 1, m, dt, o);
        } else {
            const float sg = i == n - 1 ? 1.f : (p_in[base + i + 1] > pi ? 1.f : (p_in[base + i + 1] < pi ? -1.f : 0.f));
            idm_step_f32(pi, vi, dp, dv, m, dt, o, sg);
        }
        if (o.collided && fault_index < 0) fault_index = i;
        p_out[base + i] = o.np; v_out[base + i] = o.nv;
        if (tp) {
            tp[i] = make_float4(o.dE[0], o.dE[1], o.dE[2], o.dE[3]);
            tp[Vp + i] = make_float4(o.dLd[0], o.dLd[1], o.dLd[2], o.dLd[3]);
        }
    }
    if (fault_index >= 0) raise_fault_m(err, DHTS_FAULT_COLLISION, 0, lane, fault_index);
}

}  // namespace dhts

using namespace dhts;

static inline bool micro_desc_ok(const dhts_micro_desc *d) {
    return d && d->n_lanes > 0 && d->capacity > 0 && d->capacity <= DHTS_MICRO_MAX_VEHICLES && d->dt > 0;
}
static inline int launch_status_m() { return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH; }

int dhts_micro_fwd_waves_override = 0;      // DHTS_OPT_MICRO_FWD_WAVES: 0 = heuristic, 1 / 2 / 4 wavefronts per lane

template <int K, int kW, bool kCompact>
static void launch_micro_fwd(const dhts_micro_desc *d, int T, const float *p, const float *v, const int32_t *count,
                             const double *params, const double *head, float *p_out, float *v_out, float *tape,
                             float *hist, dhts_error *err, hipStream_t s) {
    const size_t lds = sizeof(float) * (kW > 1 ? 4 : 2) * (size_t)(d->capacity + 1);
    if (count == nullptr && d->capacity == 64 * kW * K)
        micro_rollout_fwd_kernel<K, kW, kCompact, true><<<d->n_lanes, 64 * kW, lds, s>>>(d->n_lanes, d->capacity, T, d->dt, p, v, count,
                                                                                 params, head, p_out, v_out, tape, hist, err);
    else
        micro_rollout_fwd_kernel<K, kW, kCompact, false><<<d->n_lanes, 64 * kW, lds, s>>>(d->n_lanes, d->capacity, T, d->dt, p, v, count,
                                                                                  params, head, p_out, v_out, tape, hist, err);
}

// wavefronts per lane of the forward kernel (1, 2 or 4)
static int micro_fwd_waves(const dhts_micro_desc *d) {
    int W = dhts_micro_fwd_waves_override;
    if (W == 0) W = (d->capacity > 64 && (long long)d->n_lanes * 2 <= 8192) ? 2 : 1;
    if (W >= 4 && d->capacity > 128) return 4;
    if (W >= 2 && d->capacity > 64) return 2;
    return 1;
}
static int micro_fwd_passes(const dhts_micro_desc *d, int W) {       // the literal K dispatch_micro_fwd instantiates
    const int K = (d->capacity + 64 * W - 1) / (64 * W);
    return K <= 1 ? 1 : (K <= 2 ? 2 : (K <= 4 ? 4 : (K <= 8 ? 8 : 16)));
}
static int micro_bwd_block(const dhts_micro_desc *d) {
    int B = (d->capacity + 63) & ~63;
    return B > 256 ? 256 : B;
}

// passes per thread for kW wavefronts per lane
template <int kW, bool kCompact>
static void dispatch_micro_fwd(const dhts_micro_desc *d, int T, const float *p, const float *v, const int32_t *count,
                               const double *params, const double *head, float *p_out, float *v_out, float *tape,
                               float *hist, dhts_error *err, hipStream_t s) {
    const int K = (d->capacity + 64 * kW - 1) / (64 * kW);
    if (K <= 1) launch_micro_fwd<1, kW, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
    else if (K <= 2) launch_micro_fwd<2, kW, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
    else if (K <= 4) launch_micro_fwd<4, kW, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
    else if (K <= 8) launch_micro_fwd<8, kW, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
    else launch_micro_fwd<16, kW, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
}

template <bool kCompact>
static int micro_fwd_launch(const dhts_micro_desc *d, int T,
                            const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                            float *p_out, float *v_out, float *tape, float *hist, dhts_error *err, void *stream) {
    if (!micro_desc_ok(d) || T < 0 || !p || !v || !params || !head || !p_out || !v_out) return DHTS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    // wavefronts per lane: one wave keeps the whole lane free of barriers; with few lanes per SIMD more waves per lane buy
    // the latency hiding back (256 CUs x 4 SIMDs x 8 waves)
    const int W = micro_fwd_waves(d);
    if (W == 4) dispatch_micro_fwd<4, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
    else if (W == 2) dispatch_micro_fwd<2, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
    else dispatch_micro_fwd<1, kCompact>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, s);
    return launch_status_m();
}
template <bool kCompact>
static int micro_bwd_launch(const dhts_micro_desc *d, int T, const float *tape, const int32_t *count,
                            const float *g_p, const float *g_v, const float *g_hist,
                            float *g_p_out, float *g_v_out, double *g_head, int fold, dhts_error *err, void *stream) {
    if (!micro_desc_ok(d) || T < 0 || (T > 0 && !tape) || !g_p || !g_v || !g_p_out || !g_v_out) return DHTS_E_INVALID;
    const size_t lds = sizeof(float) * 4 * (size_t)(d->capacity + 2);
    const int B = micro_bwd_block(d);
    micro_rollout_bwd_kernel<kCompact><<<d->n_lanes, B, lds, (hipStream_t)stream>>>(
        d->n_lanes, d->capacity, T, d->dt, tape, count, g_p, g_v, g_hist, g_p_out, g_v_out,
        g_head, fold, err);
    return launch_status_m();
}

extern "C" {

int dhts_idm_batch(int64_t n, int variant, const double *in, double *next_pv, float *dEgo, float *dLeading, int32_t *collided,
                   double *acc_sstar, int32_t *clips, void *stream) {
    if (n < 0 || (variant != 0 && variant != 1) || !in || !next_pv || !dEgo || !dLeading || !collided || !acc_sstar || !clips) return DHTS_E_INVALID;
    if (n == 0) return DHTS_OK;
    int64_t g = (n + 255) / 256;
    idm_batch_kernel<<<(int)(g > 2048 ? 2048 : g), 256, 0, (hipStream_t)stream>>>(n, variant, in, next_pv, dEgo, dLeading, collided, acc_sstar, clips);
    return launch_status_m();
}

int dhts_idm_jac_batch(int64_t n, const double *in, float *dEgo, float *dLeading, void *stream) {
    if (n <= 0 || !in || !dEgo || !dLeading) return DHTS_E_INVALID;
    const int64_t blocks = (n + 255) / 256;
    idm_jac_batch_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, (hipStream_t)stream>>>(n, in, dEgo, dLeading);
    return launch_status_m();
}

size_t dhts_micro_tape_bytes(const dhts_micro_desc *d, int T) {
    if (!micro_desc_ok(d) || T < 0) return 0;
    return (size_t)T * d->n_lanes * ((d->capacity + 63) & ~63) * sizeof(MicroTape3);
}
size_t dhts_micro_step_tape_bytes(const dhts_micro_desc *d) {
    if (!micro_desc_ok(d)) return 0;
    return (size_t)d->n_lanes * 2 * ((d->capacity + 63) & ~63) * sizeof(float4);
}

int dhts_micro_rollout_fwd(const dhts_micro_desc *d, int T,
                           const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                           float *p_out, float *v_out, float *tape, float *hist, dhts_error *err, void *stream) {
    return micro_fwd_launch<true>(d, T, p, v, count, params, head, p_out, v_out, tape, hist, err, stream);
}
int dhts_micro_rollout_bwd(const dhts_micro_desc *d, int T, const float *tape, const int32_t *count,
                                      const float *g_p, const float *g_v, const float *g_hist,
                                      float *g_p_out, float *g_v_out, double *g_head, dhts_error *err, void *stream) {
    return micro_bwd_launch<true>(d, T, tape, count, g_p, g_v, g_hist, g_p_out, g_v_out, g_head, 1, err, stream);
}
// which kernel instantiations dhts_micro_rollout_fwd / _bwd launch for this shape: the very functions the launches call
int dhts_micro_rollout_plan(const dhts_micro_desc *d, int T, int has_count, int32_t plan[8]) {
    if (!micro_desc_ok(d) || T < 0 || !plan) return DHTS_E_INVALID;
    for (int k = 0; k < 8; ++k) plan[k] = 0;
    const int W = micro_fwd_waves(d), K = micro_fwd_passes(d, W), B = micro_bwd_block(d);
    plan[0] = W;
    plan[1] = K;
    plan[2] = (!has_count && d->capacity == 64 * W * K) ? 1 : 0;
    plan[3] = (d->capacity <= B && T > 0) ? 1 : 0;
    plan[4] = B;
    return DHTS_OK;
}
// the single-step operator keeps the reference's dqs[a][2][2][2] (32 B per vehicle)
int dhts_micro_step_fwd(const dhts_micro_desc *d,
                        const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                        float *p_out, float *v_out, float *tape, dhts_error *err, void *stream) {
    return micro_fwd_launch<false>(d, 1, p, v, count, params, head, p_out, v_out, tape, nullptr, err, stream);
}
int dhts_micro_step_fwd_tensor(const dhts_micro_desc *d,
                               const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                               float *p_out, float *v_out, float *tape, dhts_error *err, void *stream) {
    if (!micro_desc_ok(d) || !p || !v || !params || !head || !p_out || !v_out) return DHTS_E_INVALID;
    const int B = d->capacity <= 64 ? 64 : (d->capacity <= 128 ? 128 : 256);
    micro_step_tensor_fwd_kernel<false><<<d->n_lanes, B, 0, (hipStream_t)stream>>>(d->n_lanes, d->capacity, d->dt, p, v, count, params, head,
                                                                                 p_out, v_out, tape, err);
    return launch_status_m();
}
int dhts_micro_step_fwd_tensor_head(const dhts_micro_desc *d,
                                    const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                                    float *p_out, float *v_out, float *tape, dhts_error *err, void *stream) {
    if (!micro_desc_ok(d) || !p || !v || !params || !head || !p_out || !v_out) return DHTS_E_INVALID;
    const int B = d->capacity <= 64 ? 64 : (d->capacity <= 128 ? 128 : 256);
    micro_step_tensor_fwd_kernel<true><<<d->n_lanes, B, 0, (hipStream_t)stream>>>(d->n_lanes, d->capacity, d->dt, p, v, count, params, head,
                                                                                p_out, v_out, tape, err);
    return launch_status_m();
}
int dhts_micro_step_bwd(const dhts_micro_desc *d, const float *tape, const int32_t *count,
                        const float *g_p, const float *g_v,
                        float *g_p_out, float *g_v_out, double *g_head, dhts_error *err, void *stream) {
    return micro_bwd_launch<false>(d, 1, tape, count, g_p, g_v, nullptr, g_p_out, g_v_out, g_head, 0, err, stream);
}

}  // extern "C"
