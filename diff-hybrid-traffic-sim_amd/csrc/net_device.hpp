// net_device.hpp -- device helpers shared by the road-network kernels (network_kernels.hip, hybrid_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"

namespace dhts {

constexpr float kSigK = 32.f;       // sigmoid constant of the signals (_env.py:928-960, _simulator.py:128)

// The exponential of a sigmoid.  `exact` (a wave-uniform flag): the double exponential rounded once instead of the device library's expf
// (1 ulp).  The reference evaluates torch.sigmoid on CPU float32 tensors; its exp, glibc's expf and a correctly rounded exp
// agree on all but a fraction of a per cent of arguments, the device's expf differs from them on many.  One ulp in one value is
// harmless -- but in itscp `micro` mode (every lane an IDM lane stepped in float32 tensor arithmetic, a hundred vehicles following
// each other for hundreds of steps) the signals and head-gap scores steer a chaotic system, and a fuzz run over random schedules saw
// such ulps grow to 1.1-1.8e-5 of the largest queue term in 4 of 2 760 episodes (profiles/r06z_fuzz_*.log; 0 of 2 760 with the
// exact form).  The kernels ask for it exactly there: dhts_hybrid_tables::micro_tensor_ladder networks, signals and head gaps of
// differentiable episodes (+3-7 % on those episodes; every other network keeps expf and its speed).
// Two forms of it.  Out of line (kInl = false, the default): its forty instructions and their registers stay out of the callers' hot
// paths -- inlined, the persistent stepwise kernel of a network that never takes it ran 6 % slower, and the 128-register instantiation
// of the fused forward kernel 40 %.  Inline (kInl = true): what the fused kernels' roomier instantiations take -- a call costs config
// 4's forward sweep 4 %.
__device__ __forceinline__ float sig_exp_exact_inl(float x) { return (float)exp((double)x); }
__device__ __attribute__((noinline)) float sig_exp_exact_call(float x) { return (float)exp((double)x); }
template <bool kInl = false> __device__ __forceinline__ float sig_exp(float x, bool exact) {
    return exact ? (kInl ? sig_exp_exact_inl(x) : sig_exp_exact_call(x)) : expf(x);
}

// dmath/operation.py:3-30
template <bool kInl = false> __device__ __forceinline__ float soft_switch(float value, float constant, bool exact = false) {
    float z = value * constant;
    z = fminf(fmaxf(z, -16.f), 16.f);
    return 1.f / (1.f + sig_exp<kInl>(-z, exact));
}
// both at once (the reverse sweeps' loss taps): one exponential and one division instead of two of each; the same values
// (outside the clamp the gradient is 0 and the switch is the clamped one, inside z is not changed by the clamp)
template <bool kInl = false>
__device__ __forceinline__ void soft_switch_both(float value, float constant, float &s, float &ds, bool exact = false) {
    const float z = value * constant;
    const float zc = fminf(fmaxf(z, -16.f), 16.f);
    s = 1.f / (1.f + sig_exp<kInl>(-zc, exact));
    ds = (z < -16.f || z > 16.f) ? 0.f : s * (1.f - s) * constant;
}
__device__ __forceinline__ float soft_switch_grad(float value, float constant) {
    const float z = value * constant;
    if (z < -16.f || z > 16.f) return 0.f;
    const float s = 1.f / (1.f + expf(-z));
    return s * (1.f - s) * constant;
}

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also waits for every outstanding global load / store
// (vmcnt(0)), which would serialise the one-step-ahead fetches of these kernels with their barriers; all data the
// phases exchange goes through LDS, and nothing a kernel writes to global memory is read back by another thread of it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix sums over the 64 lanes of a wavefront with DPP moves only (no LDS crossbar round trips): row_shr 1/2/4/8
// inside the rows of 16 lanes, then row_bcast15 into rows 1 and 3 and row_bcast31 into rows 2 and 3.  Lanes without a DPP
// source add the identity.
namespace dpp {
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118, kBcast15 = 0x142, kBcast31 = 0x143;
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int mov0(int x) { return __builtin_amdgcn_update_dpp(0, x, kCtrl, kRowMask, 0xF, false); }
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double mov0(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = mov0<kCtrl, kRowMask>((int)(b & 0xffffffffll)), hi = mov0<kCtrl, kRowMask>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
}  // namespace dpp
template <typename T>
__device__ __forceinline__ T wave_scan_add(T x) {
    x += dpp::mov0<dpp::kRowShr1, 0xF>(x);
    x += dpp::mov0<dpp::kRowShr2, 0xF>(x);
    x += dpp::mov0<dpp::kRowShr4, 0xF>(x);
    x += dpp::mov0<dpp::kRowShr8, 0xF>(x);
    x += dpp::mov0<dpp::kBcast15, 0xA>(x);
    x += dpp::mov0<dpp::kBcast31, 0xC>(x);
    return x;
}
// inclusive prefix sum inside each row of 16 lanes (lane 15 of a row ends up with the row's total)
template <typename T>
__device__ __forceinline__ T row_scan_add(T x) {
    x += dpp::mov0<dpp::kRowShr1, 0xF>(x);
    x += dpp::mov0<dpp::kRowShr2, 0xF>(x);
    x += dpp::mov0<dpp::kRowShr4, 0xF>(x);
    x += dpp::mov0<dpp::kRowShr8, 0xF>(x);
    return x;
}
__device__ __forceinline__ int wave_last(int x) { return __builtin_amdgcn_readlane(x, 63); }
__device__ __forceinline__ double wave_last(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

struct NetTables {
    const int32_t *lane_ncell, *lane_off, *sig_kind, *inter;   // [L]
    const double *lane_dx;                                      // [L]
    const int32_t *left_src, *left_gate, *right_src;            // [T][L] (per replica when table_stride != 0)
    const double *schedule;                                     // [T][L]
    size_t table_stride;                                        // elements between replicas in the [T][L] tables (0 = shared)
    const int32_t *nxt_ptr, *nxt_idx, *prv_ptr, *prv_idx;       // static adjacency, CSR, ascending ids
    int n_edges;
};

__device__ __forceinline__ void net_fault(dhts_error *err, int code, int step, int lane, int index) {
    if (err == nullptr) return;
    if (atomicCAS(&err->code, 0, code) == 0) { err->step = step; err->lane = lane; err->index = index; }
}

// phase signals of intersection k at step t: west-east and north-south switches and their inputs (_env.py:885-962).
// `phase_raw` = t / F and `frame` = t % F are passed in so that rollouts can count them instead of dividing every step.
// hard = an evaluation episode (differentiable = False): float(a > progress), float(progress > a) (_env.py:928-960).
template <bool kInl = false>
__device__ __forceinline__ void phase_signal_at(const float *action, int n_action, int sq, int F, int phase_raw, int frame, int k,
                                                float &we, float &ns, float &a, float &prog, int &a_index, bool hard = false,
                                                bool exact = false) {
    const int last = n_action / sq - 1;
    const int phase = phase_raw > last ? last : phase_raw;
    double pr = (double)frame / (double)F;
    pr = pr > 1.0 ? 1.0 : pr;
    a_index = phase * sq + k;
    a = action[a_index];
    prog = (float)pr;
    if (hard) { we = a > prog ? 1.f : 0.f; ns = prog > a ? 1.f : 0.f; return; }
    we = soft_switch<kInl>(a - prog, kSigK, exact);
    ns = soft_switch<kInl>(prog - a, kSigK, exact);
}
__device__ __forceinline__ void phase_signal(const float *action, int n_action, int sq, int F, int t, int k,
                                             float &we, float &ns, float &a, float &prog, int &a_index) {
    phase_signal_at(action, n_action, sq, F, t / F, t % F, k, we, ns, a, prog, a_index);
}

}  // namespace dhts
