// fast_math.hpp -- double-precision helpers built on the gfx950 v_rsq_f64 / v_rcp_f64 seeds (2^-23 relative) plus
// Newton / Goldschmidt refinement with explicit fma.  Arguments must be in the normal range (callers pass >= 1e-5).
#pragma once
#include <hip/hip_runtime.h>

namespace dhts {

// s = sqrt(x), h = 0.5 / sqrt(x) for x in the normal range (callers pass x >= 1e-5): v_rsq_f64 seed (2^-23),
// one Goldschmidt step, two residual corrections of s (the sequence LLVM uses for a correctly rounded f64 sqrt,
// without its denormal scaling) and one of h.  Both results are within ~1 ulp.
__device__ __forceinline__ void sqrt_hrsqrt(double x, double &s, double &h) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    s = __builtin_fma(d, h, g);
    r = __builtin_fma(-h, s, 0.5);
    h = __builtin_fma(h + h, r, h);
}
__device__ __forceinline__ double fast_sqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// 1 / x: v_rcp_f64 seed + two Newton steps (~1 ulp)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

}  // namespace dhts
