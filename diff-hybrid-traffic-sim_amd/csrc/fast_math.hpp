// fast_math.hpp -- double-precision helpers built on the gfx950 v_rsq_f64 / v_rcp_f64 seeds (2^-23 relative) plus
// Newton / Goldschmidt refinement with explicit fma.  Arguments must be in the normal range (callers pass >= 1e-5).
#pragma once
#include <hip/hip_runtime.h>

namespace dhts {

// s = sqrt(x), h = 0.5 / sqrt(x): v_rsq_f64 seed (relative error <= 2^-23) + ONE coupled Goldschmidt step, which squares
// the error: both results are good to ~2e-14 relative.  That is 6 orders below half a float32 ulp, so a float32 value
// rounded from an expression built on them differs from the exactly rounded one with probability ~1e-6; the exact
// IEEE sequences (13 / 10 dependent double operations instead of 6) are kept in the *_ieee reference-order variants.
__device__ __forceinline__ void sqrt_hrsqrt(double x, double &s, double &h) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    const double h0 = 0.5 * y;
    const double r = __builtin_fma(-h0, g, 0.5);
    s = __builtin_fma(g, r, g);
    h = __builtin_fma(h0, r, h0);
}
__device__ __forceinline__ double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    const double h0 = 0.5 * y;
    const double r = __builtin_fma(-h0, g, 0.5);
    return __builtin_fma(g, r, g);
}

// 1 / x: v_rcp_f64 seed (<= 2^-23) + one Newton step (~1e-14 relative)
__device__ __forceinline__ double fast_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

}  // namespace dhts
