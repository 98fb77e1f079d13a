// dhts_common.hip -- library-level entry points of the C ABI (include/dhts.h) and the one kernel every network path shares: the
// reward in the reference's own summation order.
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"

int dhts_opt_reward_chain = 0;      // DHTS_OPT_REWARD_CHAIN (dhts_set_option, macro_kernels.hip)

namespace dhts {

// ItscpEnv._reward (example/control/itscp/_env.py:770-797): `reward = 0; for lane: for x in queue_length[lane]: reward = reward +
// (-1.0) * x` -- ONE running sum, lanes outermost.  In a differentiable episode every x is a float32 tensor, so the sum is a
// float32 chain of L * T dependent additions from the first term on.  In an evaluation episode an IDM lane's term is the Python
// float (n ** 2.0) * dt (_env.py:709-738) and a cell lane's a float32 tensor: the sum is a double until the first cell lane's
// term joins it and a float32 tensor from then on (an IDM term is cast when it meets the tensor); the float32 queue array holds
// the rounded terms, and the double ones are recovered from it exactly (n is a small whole number).
// The rollout kernels themselves add a lane's steps first and the lanes' subtotals last (a thread per lane): the same real
// number, another float32 rounding (config 4: 6e-6 relative).  This kernel is the reference's chain: one wavefront per replica,
// 64 terms fetched at a time (one per lane of the wavefront), then added one after the other from a scalar register.
// queue [R][T][L]; reward[r * stride] <- the chain; stride == 2: reward[r * 2 + 1] <- the chain over the first `cut` steps.
__global__ void __launch_bounds__(64) reward_chain_kernel(int T, int L, const float *__restrict__ queue, const int32_t *__restrict__ lane_macro,
                                                          int hard, double dt, int cut, float *__restrict__ reward, int stride) {
    const int rep = blockIdx.x, ln = threadIdx.x;
    const float *q = queue + (size_t)rep * T * L;
    float rf = 0.f, rc = 0.f;
    double rd = 0.;
    bool tensor = hard == 0;
    for (int l = 0; l < L; ++l) {
        const bool macro = lane_macro == nullptr || lane_macro[l] != 0;
        if (macro && !tensor) { rf = (float)rd; tensor = true; }
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int n = T - t0 < 64 ? T - t0 : 64;
            const float v = ln < n ? q[(size_t)(t0 + ln) * L + l] : 0.f;
            if (tensor) {
                if (n == 64 && t0 + 64 <= cut) {
#pragma unroll
                    for (int i = 0; i < 64; ++i) {
                        const float x = (-1.0f) * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
                        rf = rf + x; rc = rc + x;
                    }
                } else {
                    for (int i = 0; i < n; ++i) {
                        const float x = (-1.0f) * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
                        rf = rf + x;
                        if (t0 + i < cut) rc = rc + x;
                    }
                }
            } else {
                for (int i = 0; i < n; ++i) {
                    const double x = (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
                    const double cnt = rint(sqrt(x / dt));
                    rd = rd + -1.0 * ((cnt * cnt) * dt);
                }
            }
        }
    }
    if (ln == 0) {
        const float r = tensor ? rf : (float)rd;
        reward[(size_t)rep * stride] = r;
        if (stride == 2) reward[(size_t)rep * 2 + 1] = hard ? r : rc;      // (an evaluation episode has no restricted reward)
    }
}

}  // namespace dhts

// host side, for the launchers of the other translation units (no kernel call crosses a file)
int dhts_launch_reward_chain(int R, int T, int L, const float *queue, const int32_t *lane_macro, int hard, double dt, int loss_steps,
                             float *reward, int stride, void *stream) {
    if (R <= 0 || T <= 0 || L <= 0) return DHTS_OK;
    const int cut = (loss_steps > 0 && loss_steps < T) ? loss_steps : T;
    dhts::reward_chain_kernel<<<R, 64, 0, (hipStream_t)stream>>>(T, L, queue, lane_macro, hard, dt, cut, reward, stride);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

extern "C" {

int dhts_version(void) { return DHTS_VERSION; }

int dhts_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DHTS_E_NO_DEVICE;
    return n;
}

}  // extern "C"
