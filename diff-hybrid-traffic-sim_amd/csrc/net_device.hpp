// net_device.hpp -- device helpers shared by the road-network kernels (network_kernels.hip, hybrid_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"

namespace dhts {

constexpr float kSigK = 32.f;       // sigmoid constant of the signals (_env.py:928-960, _simulator.py:128)

// dmath/operation.py:3-30
__device__ __forceinline__ float soft_switch(float value, float constant) {
    float z = value * constant;
    z = fminf(fmaxf(z, -16.f), 16.f);
    return 1.f / (1.f + expf(-z));
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

// phase signals of intersection k at step t: west-east and north-south switches and their inputs (_env.py:885-962)
__device__ __forceinline__ void phase_signal(const float *action, int n_action, int sq, int F, int t, int k,
                                             float &we, float &ns, float &a, float &prog, int &a_index) {
    int phase = t / F;
    const int last = n_action / sq - 1;
    phase = phase > last ? last : phase;
    double pr = (double)(t % F) / (double)F;
    pr = pr > 1.0 ? 1.0 : pr;
    a_index = phase * sq + k;
    a = action[a_index];
    prog = (float)pr;
    we = soft_switch(a - prog, kSigK);
    ns = soft_switch(prog - a, kSigK);
}

}  // namespace dhts
