// arz_device.hpp -- device math of the ARZ cell stencil (gfx950).
//
// Implements, per cell interface, what the reference computes in model/macro/_arz.py:212-332 (exact Riemann
// solver), model/macro/darz.py:12-233 (analytic Jacobians of Q_0 and of the flux) and the 2x2 products of
// road/lane/dmacro_lane.py:115-129 -- and, per cell, the float32 state glue of _arz.py:82-92.
//
// Precision ladder (SURVEY.md 8a Note P): inputs are float32 state widened to double, the solve and every
// Jacobian entry are double, entries are rounded to float32, 2x2 products and the adjoint are float32 with
// NumPy's accumulation form acc = a0*b0; acc = fma(a1, b1, acc).  The translation unit is compiled with
// -ffp-contract=off so nothing else is fused.  gamma = 0.5 is a compile-time constant: x**gamma is sqrt(x),
// x**(1/gamma) is x*x, x**(gamma-1) is 1/sqrt(x) (the reference's only exponents).
#pragma once
#include <hip/hip_runtime.h>

#include "fast_math.hpp"

namespace dhts {

constexpr double kEps = 1e-5;             // EPSILON, _arz.py:2
constexpr float kEpsF = 1e-5f;            // the same constant cast to float32 where it meets a tensor
constexpr double kG = 0.5;                // GAMMA, _arz.py:1
constexpr double kG1 = kG + 1.0;          // gamma + 1
constexpr double kGoG1 = kG / (kG + 1.0); // gamma / (gamma + 1)
constexpr double kHalfRsqrtEps = 158.11388300841895;   // 0.5 / sqrt(1e-5) = gamma * EPSILON ** (gamma - 1)

// Python's max(a, b): b if b > a else a
__device__ __forceinline__ double pymax(double a, double b) { return (b > a) ? b : a; }

// compute_u_eq, _arz.py:133-138
__device__ __forceinline__ double u_eq_d(double r, double um) {
    r = pymax(r, 0.);
    return um * (1. - sqrt(r + kEps));
}
// compute_u_eq_prime, _arz.py:146-149
__device__ __forceinline__ double u_eq_prime_d(double r, double um) {
    r = pymax(r, kEps);
    return -um * kG * (1.0 / sqrt(r));
}

// ---- float32 glue (0-dim torch tensors in the reference) ---------------------------------------------
__device__ __forceinline__ float glue_u_eq(float r, float um) {
    if (0.f > r) return (float)((double)um * (1. - sqrt(0. + kEps)));   // max(r, 0.) picked the Python float
    float t = r + kEpsF;
    t = sqrtf(t);
    t = 1.f - t;
    return um * t;
}
// FullQ.set_r_u / from_r_u, _arz.py:73-86
__device__ __forceinline__ void glue_from_r_u(float r, float u, float um, float &y, float &ueq) {
    ueq = glue_u_eq(r, um);
    y = r * (u - ueq);
}
// The upstream ghost of an itscp SOURCE lane is a cell of Python floats in the reference (_simulator.py:68-71: r = the inflow from the
// schedule, u = u_eq(r), both double; dMacroLane.decell leaves plain floats alone) and its Riemann solve reads them as such.  Through a
// float32 ghost quad (r, y, u, u_eq) it travels as {NaN, 0, low word of r, high word of r}: a NaN density marks it (no ordinary ghost
// has one), u = u_eq = u_eq(r) is evaluated in double where the quad is read, y = r (u - u_eq(r)) = 0.
__device__ __forceinline__ void ghost_source_pack(double r, float &g0, float &g1, float &g2, float &g3) {
    g0 = __int_as_float(0x7fc00000); g1 = 0.f;
    g2 = __int_as_float(__double2loint(r)); g3 = __int_as_float(__double2hiint(r));
}
__device__ __forceinline__ bool ghost_is_source(float g0) { return g0 != g0; }
__device__ __forceinline__ void ghost_source_unpack(float g2, float g3, double um, double &r, double &y, double &u, double &ueq) {
    r = __hiloint2double(__float_as_int(g3), __float_as_int(g2));
    u = u_eq_d(r, um);
    y = 0.; ueq = u;
}
// FullQ.set_r_y, _arz.py:88-92
__device__ __forceinline__ void glue_from_r_y(float r, float y, float um, float &u, float &ueq) {
    ueq = glue_u_eq(r, um);
    if (r < kEpsF) {
        double ueq_c = (double)um * (1. - sqrt(kEps + kEps));       // u_eq(1e-5) in Python floats
        u = ((y) / (kEpsF)) + (float)ueq_c;
    } else {
        u = ((y) / (r)) + ueq;
    }
}
// adjoint of u = y / rc + u_eq(rc) as torch autograd evaluates it (float32)
__device__ __forceinline__ void glue_u_bwd(float r, float y, float um, float g_u, float &g_r, float &g_y) {
    if (r < kEpsF) {
        g_y += ((g_u) / (kEpsF));
        return;
    }
    g_y += ((g_u) / (r));
    float gd = -g_u * ((((y) / (r))) / (r));
    float gp = 0.f;
    if (!(0.f > r)) {
        float t = r + kEpsF;
        float gs = -(g_u * um);
        gp = gs * (0.5f * ((1.f) / (sqrtf(t))));
    }
    g_r += gd + gp;
}
// adjoint of y = r * (u - u_eq(r)); ACCUMULATES into g_r and g_u (callers may already hold a cotangent of u, e.g. the
// loss tap on a deposited cell's speed or the stored-ghost chain)
__device__ __forceinline__ void glue_y_bwd(float r, float u, float um, float g_y, float &g_r, float &g_u) {
    float ueq = glue_u_eq(r, um);
    float diff = u - ueq;
    float g_diff = g_y * r;
    g_u += g_diff;
    float acc = g_y * diff;
    if (!(0.f > r)) {
        float t = r + kEpsF;
        float gs = -((-g_diff) * um);
        acc += gs * (0.5f * ((1.f) / (sqrtf(t))));
    }
    g_r += acc;
}

// float32 2-term dot the way NumPy's matmul accumulates it
__device__ __forceinline__ float dot2(float a0, float b0, float a1, float b1) { return __fmaf_rn(a1, b1, a0 * b0); }

struct Iface {
    double Fr, Fy;   // flux of Q_0: (r*u, y*u)                                  (_arz.py:94-101)
    float A[4];      // fp(Q_0) @ dQ_0/dQ_L   row-major 2x2, float32             (dmacro_lane.py:126-129)
    float B[4];      // fp(Q_0) @ dQ_0/dQ_R
    bool cfl_bad;    // dt >= dx / max(|speed|, 1e-5) for speed0 or speed1       (_macro_lane.py:141-146)
};

struct IfaceDebug {  // intermediate results exported only by the known-answer entry point (dhts_arz_interface_batch)
    int ci;          // case_ind: 0 = Q_L, 1 = Q_M, 2 = Q_C
    double q0[4];    // Q_0 = (r, y, u, u_eq)
    float dL[4], dR[4], fp[4];
    double speed[2]; // speed0, speed1 of ARZ.riemann_solve (_arz.py:222-314, returned :316-332)
};

struct IfaceConst {  // per-launch constants
    double um, inv_um, inv_15um, dt, dx;
    double cfl_lim, cfl_lim2;   // dx / dt and twice that (production path: |speed| < dx / dt instead of dt * |speed| < dx)
    double ueq_2eps;            // u_eq(eps) = um (1 - sqrt(eps + eps)): u_eq of a clamped left density (_arz.py:155-165)
    double inv_epsf;            // 1 / float32(eps): the float32 glue divides by max(r, eps) with eps cast to float32
    float ueq_2eps_f, ueq_neg_f;   // float32 casts of u_eq(eps) and of u_eq(0) = um (1 - sqrt(eps)) (negative density), as the glue uses them
    bool lim_ok;                // 1e-5 < dx / dt: the max(|speed|, 1e-5) floor of the CFL assert
    __host__ __device__ void set_um(double um_) {
        um = um_; inv_um = 1.0 / um_; inv_15um = 1.0 / (1.5 * um_); ueq_2eps = um_ * (1. - sqrt(1e-5 + 1e-5));
        inv_epsf = 1.0 / (double)1e-5f;
        const float umf = (float)um_;               // the glue works with the float32 speed limit
        ueq_2eps_f = (float)((double)umf * (1. - sqrt(1e-5 + 1e-5)));
        ueq_neg_f = (float)((double)umf * (1. - sqrt(0. + 1e-5)));
    }
    __host__ __device__ void set_grid(double dt_, double dx_) {
        dt = dt_; dx = dx_; cfl_lim = dx_ / dt_; cfl_lim2 = 2.0 * cfl_lim; lim_ok = 1e-5 < cfl_lim;
    }
};

// The two wave speeds ARZ.riemann_solve reports (_arz.py:222-314), in the reference's own order of operations (IEEE division
// and square root).  The rollout kernels only need their CFL test -- the production solve never forms the shock speed's
// quotient -- so this runs for the known-answer entry point alone (dhts_arz_interface_batch, the mirror's ARZ.riemann_solve).
__device__ __forceinline__ void arz_speeds_ref(double rL, double uL, double qL, double rR, double uR, double um, double &s0, double &s1) {
    const double ueqp_L = u_eq_prime_d(rL, um);
    if (rL < kEps) { s0 = 0.0; s1 = uL; return; }
    const double l0l = uL + rL * ueqp_L;
    const double qm_u = um + uL - qL;
    if (rR < kEps) { s0 = (l0l + qm_u) * 0.5; s1 = s0; return; }
    if (fabs(uL - uR) < kEps) { s0 = 0.0; s1 = uR; return; }
    const double b = sqrt(rL) + ((uL - uR) / um);
    const double rm = b * b;
    if (uL > uR) { s0 = (rm * uR - rL * uL) / pymax(rm - rL, kEps); s1 = uR; return; }
    if (qm_u > uR) { s0 = (l0l + (uR + rm * u_eq_prime_d(rm, um))) * 0.5; s1 = uR; return; }
    s0 = (l0l + qm_u) * 0.5; s1 = uR;
}

// Reference-order version: IEEE double division and square root exactly where the reference divides and takes
// powers.  Selected with -DDHTS_IEEE_DIV_SQRT (validation builds); the default build uses arz_interface_fast.
__device__ __forceinline__ void arz_interface_ieee(double rL, double yL, double uL, double qL,
                                                   double rR, double yR, double uR, double qR,
                                                   const IfaceConst &k, Iface &o, IfaceDebug *dbg = nullptr) {
    const double um = k.um;
    // ---- Riemann solve: case index and speeds (_arz.py:222-314) ----
    int ci;
    double s0, s1;
    double rm = 0., l0m_u = 0.;
    const double sqrt_rL = sqrt(rL);                       // r_L ** gamma
    const double ueqp_L = u_eq_prime_d(rL, um);
    if (rL < kEps) {
        s0 = 0.0; s1 = uL; ci = 0;
    } else if (rR < kEps) {
        double qm_u = um + uL - qL;
        double l0l = uL + rL * ueqp_L;
        s0 = (l0l + qm_u) * 0.5; s1 = s0;
        ci = (l0l >= 0.0) ? 0 : 2;
    } else if (fabs(uL - uR) < kEps) {
        s0 = 0.0; s1 = uR; ci = 0;
    } else if (uL > uR) {
        double b = sqrt_rL + ((uL - uR) / um);
        rm = b * b;                                        // compute_Qm :194
        double diff = rm * uR - rL * uL;
        s0 = diff / pymax(rm - rL, kEps); s1 = uR;
        ci = (s0 >= 0.0) ? 0 : 1;
    } else if (um + uL - qL > uR) {
        double b = sqrt_rL + ((uL - uR) / um);
        rm = b * b;
        double l0l = uL + rL * ueqp_L;
        double l0m = uR + rm * u_eq_prime_d(rm, um);
        s0 = (l0l + l0m) * 0.5; s1 = uR;
        ci = (l0l >= 0) ? 0 : ((l0m <= 0) ? 1 : 2);
    } else {
        double qm_u = um + uL - qL;
        double l0l = uL + rL * ueqp_L;
        s0 = (l0l + qm_u) * 0.5; s1 = uR;
        ci = (l0l >= 0.0) ? 0 : 2;
    }
    (void)l0m_u;
    o.cfl_bad = !(k.dt * pymax(fmax(fabs(s0), fabs(s1)), 1e-5) < k.dx);

    // ---- Q_0 (_arz.py:155-199, 316-326), its Jacobians (darz.py:12-192) ----
    double r0, y0, u0, q0;
    float dL[4], dR[4];
    const double rLc = pymax(rL, kEps);
    if (ci == 0) {
        // compute_Ql: set_r_y on Python floats
        r0 = rL; y0 = yL;
        u0 = (yL / rLc) + u_eq_d(rLc, um);
        q0 = u_eq_d(rL, um);
    } else if (ci == 1) {
        r0 = rm; u0 = uR;
        q0 = u_eq_d(rm, um);
        y0 = rm * (uR - q0);
        // compute_dM, darz.py:35-122
        const double rRc = pymax(rR, kEps);
        const double ueqp_M = u_eq_prime_d(rm, um);
        const double duL_drL = -yL / (rLc * rLc) + ueqp_L;
        const double duL_dyL = 1.0 / rLc;
        const double duR_drR = -yR / (rRc * rRc) + u_eq_prime_d(rRc, um);
        const double duR_dyR = 1.0 / rRc;
        const double a = (1.0 / kG) * sqrt(rm);
        const double b = kG * (1.0 / sqrt(rLc));
        const double c = (1.0 / um) * duL_drL;
        const double drM_drL = a * (b + c);
        const double d = (1.0 / um) * duL_dyL;
        const double drM_dyL = a * d;
        const double e = u0 - q0;
        const double dyM_drL = drM_drL * e + rm * (-ueqp_M * drM_drL);
        const double dyM_dyL = drM_dyL * e + rm * (-ueqp_M * drM_dyL);
        const double f = (-1.0 / um) * duR_drR;
        const double drM_drR = a * f;
        const double g = (-1.0 / um) * duR_dyR;
        const double drM_dyR = a * g;
        const double dyM_drR = drM_drR * e + rm * (duR_drR - ueqp_M * drM_drR);
        const double dyM_dyR = drM_dyR * e + rm * (duR_dyR - ueqp_M * drM_dyR);
        dL[0] = (float)drM_drL; dL[1] = (float)drM_dyL; dL[2] = (float)dyM_drL; dL[3] = (float)dyM_dyL;
        dR[0] = (float)drM_drR; dR[1] = (float)drM_dyR; dR[2] = (float)dyM_drR; dR[3] = (float)dyM_dyR;
    } else {
        // compute_Qc, _arz.py:167-182
        const double base = uL + um * sqrt_rL;
        const double t = base / (kG1 * um);
        r0 = t * t;
        u0 = kGoG1 * base;
        q0 = u_eq_d(r0, um);
        y0 = r0 * (u0 - q0);
        // compute_dC, darz.py:124-192
        const double ueqp_C = u_eq_prime_d(r0, um);
        const double duL_drL = -yL / (rLc * rLc) + ueqp_L;
        const double duL_dyL = 1.0 / rLc;
        const double f = um * kG * (1.0 / sqrt(rLc));
        const double duC_drL = kGoG1 * (duL_drL + f);
        const double duC_dyL = kGoG1 * duL_dyL;
        const double b = kG1 * um;
        const double c = sqrt(r0);
        const double d = c / kG;
        const double e = d / b;
        const double drC_drL = e * (duL_drL + f);
        const double drC_dyL = e * duL_dyL;
        const double g = u0 - q0;
        const double dyC_drL = drC_drL * g + r0 * (duC_drL - ueqp_C * drC_drL);
        const double dyC_dyL = drC_dyL * g + r0 * (duC_dyL - ueqp_C * drC_dyL);
        dL[0] = (float)drC_drL; dL[1] = (float)drC_dyL; dL[2] = (float)dyC_drL; dL[3] = (float)dyC_dyL;
    }
    o.Fr = r0 * u0;
    o.Fy = y0 * u0;

    // ---- flux Jacobian at Q_0 (darz.py:217-233), float32 entries ----
    const double r0c = pymax(r0, kEps);
    const double ueqp_0 = u_eq_prime_d(r0c, um);
    const double yor = y0 / r0c;
    float fp[4];
    fp[0] = (float)(q0 + r0c * ueqp_0);
    fp[1] = 1.f;
    fp[2] = (float)(y0 * ueqp_0 - yor * yor);
    fp[3] = (float)((2.0 * y0) / r0c + q0);

    if (dbg) {
        dbg->ci = ci;
        dbg->q0[0] = r0; dbg->q0[1] = y0; dbg->q0[2] = u0; dbg->q0[3] = q0;
        dbg->speed[0] = s0; dbg->speed[1] = s1;
        for (int j = 0; j < 4; ++j) {
            dbg->fp[j] = fp[j];
            dbg->dL[j] = (ci == 0) ? ((j == 0 || j == 3) ? 1.f : 0.f) : dL[j];
            dbg->dR[j] = (ci == 1) ? dR[j] : 0.f;
        }
    }
    // ---- fp @ dL, fp @ dR in float32 (np.matmul) ----
    if (ci == 0) {          // dL = I, dR = 0: the products are fp and 0 exactly
        o.A[0] = fp[0]; o.A[1] = fp[1]; o.A[2] = fp[2]; o.A[3] = fp[3];
        o.B[0] = o.B[1] = o.B[2] = o.B[3] = 0.f;
    } else {
        o.A[0] = dot2(fp[0], dL[0], fp[1], dL[2]);
        o.A[1] = dot2(fp[0], dL[1], fp[1], dL[3]);
        o.A[2] = dot2(fp[2], dL[0], fp[3], dL[2]);
        o.A[3] = dot2(fp[2], dL[1], fp[3], dL[3]);
        if (ci == 1) {
            o.B[0] = dot2(fp[0], dR[0], fp[1], dR[2]);
            o.B[1] = dot2(fp[0], dR[1], fp[1], dR[3]);
            o.B[2] = dot2(fp[2], dR[0], fp[3], dR[2]);
            o.B[3] = dot2(fp[2], dR[1], fp[3], dR[3]);
        } else {
            o.B[0] = o.B[1] = o.B[2] = o.B[3] = 0.f;
        }
    }
}

// Production version of the interface solve.  Same formulas and branch structure as arz_interface_ieee, with
//   * every 1/sqrt(x), 1/x and sqrt(x) of one argument taken from ONE rsq + Goldschmidt sequence,
//   * divisions by launch constants turned into multiplications by their reciprocals,
//   * the shock speed's division removed (only its sign and its CFL bound are used),
//   * a*b+c written as an explicit fma where it is one (the translation unit is compiled with contraction off, so the
//     result of an interface does not depend on which kernel, or which phase of a kernel, evaluates it).
// It comes in three pieces so that a kernel can run the cheap part on every interface and the rest only where needed:
//   arz_pre_fast      left-cell quantities, case index, CFL flag                       (_arz.py:222-314)
//   arz_trivial_fast  Q_0 = Q_L (case 0: 89 % of the interfaces of BASELINE config 2), flux, flux Jacobian
//   arz_interface_fast = arz_pre_fast + (trivial | Q_M / Q_C with their Jacobians) + the float32 2x2 products
// Results differ from the reference-order version by a few double ulps, i.e. by < 1e-8 of a float32 ulp
// before the float32 stores (tests/test_gpu_parity.py checks the result against the golden vectors).
// What a cell contributes to the interface on its right, as a function of its density alone: computed once by whoever
// owns the cell (the two-phase rollout kernel keeps it beside the state in LDS), or on the spot by arz_pre_fast.
struct CellPre {
    double s, h;     // sqrt(max(r, eps)) and 0.5 / that
    double q0;       // u_eq(r) in double = um (1 - sqrt(max(r, 0) + eps))     (compute_u_eq, _arz.py:133-138)
};
__device__ __forceinline__ void arz_cell_pre(double r, double um, CellPre &c) {
    sqrt_hrsqrt(fmax(r, kEps), c.s, c.h);
    c.q0 = __builtin_fma(-um, fast_sqrt(fmax(r, 0.) + kEps), um);
}

struct IfacePre {
    double rLc, sL, hL, inv_rLc, ueqp_L, q0L;   // max(r_L, eps), its sqrt, 0.5 / sqrt, 1 / it, u_eq'(r_L), u_eq(r_L) in double
    double dU, bm;                               // u_L - u_R, sqrt(r_m) up to sign
    int ci;                                      // case_ind: 0 = Q_L, 1 = Q_M, 2 = Q_C
    bool vacL, cfl_bad;
};

__device__ __forceinline__ void arz_classify_fast(double rL, double uL, double qL, double rR, double uR, const CellPre &cl,
                                                  const IfaceConst &k, IfacePre &p) {
    const double um = k.um;
    p.rLc = fmax(rL, kEps);
    p.sL = cl.s; p.hL = cl.h; p.q0L = cl.q0;               // sL = sqrt(rLc); 2 hL = rLc^-1/2
    const double rsL = p.hL + p.hL;
    p.inv_rLc = rsL * rsL;                                 // 1 / rLc
    p.ueqp_L = -um * p.hL;                                 // u_eq'(rL) = -um * gamma * rLc^(gamma-1)
    // ---- Riemann solve: case index, CFL flag (_arz.py:222-314), evaluated branch-free: the six branches only
    //      differ in a handful of cheap candidates, which are all computed and then selected ----
    const bool vacL = rL < kEps;
    const bool vacR = rR < kEps;
    p.vacL = vacL;
    p.dU = uL - uR;
    const bool same = fabs(p.dU) < kEps;
    const bool wave_m = !(vacL | vacR | same);             // branches 4, 5, 6
    const bool b4 = wave_m & (uL > uR);
    const double qm_u = um + uL - qL;                      // u of Q_m = (0, .) next to vacuum (:235, :301)
    const bool b5 = wave_m & !b4 & (qm_u > uR);
    p.bm = __builtin_fma(p.dU, k.inv_um, p.sL);            // sqrt(r_m) up to sign (sqrt(rL) = sL when rL >= eps)
    const double rm = p.bm * p.bm;                         // compute_Qm :194 (used by branches 4 and 5 only)
    const double abm = fabs(p.bm);
    const double l0l = __builtin_fma(rL, p.ueqp_L, uL);
    const double diff = __builtin_fma(rm, uR, -(rL * uL)); // numerator of the shock speed (:263-265)
    const double den = fmax(rm - rL, kEps);
    // lambda_0(Q_m) = u_R + r_m u_eq'(r_m) = u_R - gamma u_max r_m^gamma = u_R - u_max |b| / 2   (r_m >= eps),
    // and u_R + r_m (-u_max gamma eps^(gamma-1)) below eps: no square root needed
    const double l0m = (rm >= kEps) ? __builtin_fma(-0.5 * um, abm, uR) : __builtin_fma(rm, -um * kHalfRsqrtEps, uR);
    const bool l0l_ok = l0l >= 0.0;
    const bool zero0 = vacL | (same & !vacR);              // branches 1 and 3: speed0 = 0
    // case 0 unless the wave that sits on x = 0 is the middle state: branch 4 decides by the sign of speed0 = diff / den,
    // branches 2, 5, 6 by lambda_0(Q_L); what replaces Q_L is Q_M in branch 4, and in branch 5 when lambda_0(Q_m) <= 0
    const bool triv = zero0 | (b4 & (diff >= 0.0)) | (!b4 & l0l_ok);
    const bool mid = b4 | (b5 & (l0m <= 0));
    p.ci = triv ? 0 : (mid ? 1 : 2);
    // CFL: dt * max(|speed|, 1e-5) < dx for speed0 and speed1  <=>  |speed| < dx / dt and 1e-5 < dx / dt.  Every speed the solver
    // can report is bounded by the sum below (see arz_is_trivial_fast): where that bound is inside the limit -- everywhere, in
    // a run that is not about to fail -- the exact tests are skipped (a wave-uniform branch; a NaN fails the bound's test and
    // then every exact one).
    const double bound = __builtin_fma(0.5, fabs(qL), fabs(uL) + fabs(uR)) + __builtin_fma(0.5 * um, p.sL, 0.5 * um);
    const bool sure = (bound < k.cfl_lim) & k.lim_ok;
    p.cfl_bad = false;
    if (__builtin_amdgcn_ballot_w64(!sure)) {
        asm volatile("" ::: "memory");                     // (keeps it a branch)
        // speed0 is 0, diff / den (den > 0: tested as |diff| < den dx / dt, no division) or the mean of two characteristic
        // speeds (tested as |sum| < 2 dx / dt); every test is a lane mask, the selection between them mask logic.
        const bool ok_b4 = fabs(diff) < k.cfl_lim * den;
        const bool ok_m = fabs(l0l + l0m) < k.cfl_lim2;        // branch 5
        const bool ok_q = fabs(l0l + qm_u) < k.cfl_lim2;       // branches 2 and 6
        const bool ok_avg = (b5 & ok_m) | (!b5 & ok_q);
        const bool ok0 = zero0 | (b4 & ok_b4) | (!b4 & ok_avg);
        const bool ok_uL = fabs(uL) < k.cfl_lim, ok_uR = fabs(uR) < k.cfl_lim;
        const bool ok1 = (vacL & ok_uL) | (!vacL & ((vacR & ok_avg) | (!vacR & ok_uR)));
        p.cfl_bad = !(ok0 & ok1 & k.lim_ok);
    }
}

__device__ __forceinline__ void arz_pre_fast(double rL, double uL, double qL, double rR, double uR, const IfaceConst &k,
                                             IfacePre &p) {
    CellPre cl;
    arz_cell_pre(rL, k.um, cl);
    arz_classify_fast(rL, uL, qL, rR, uR, cl, k, p);
}

// The hot-path subset of arz_classify_fast: is the interface trivial (case 0) AND provably inside the CFL bound?
// Every speed the solver can report is bounded by |u_L| + |u_R| + |u_eq,L| / 2 + u_max (1 + sqrt(r_L)) / 2 (shock:
// |u_R| + u_max sqrt(r_L) / 2 with or without the eps clamp of its denominator; lambda_0(Q_L): |u_L| + u_max sqrt(r_L) / 2;
// lambda_0(Q_m): |u_R| + u_max |b| / 2 with |b| <= sqrt(r_L) + |u_L - u_R| / u_max; u of the vacuum-side state:
// u_max + |u_L| + |u_eq,L|).  An interface that fails the bound is treated as non-trivial, i.e. handed to the full solver
// with its exact CFL test.  Fills the fields of `p` that arz_trivial_fast reads.
// (vacR = r_R < eps is passed in: a caller that holds r_R as float32 tests it there -- r_R < 1e-5 (double) <=> r_R <= float32(1e-5),
// because float32(1e-5) < 1e-5 < its successor)
__device__ __forceinline__ bool arz_is_trivial_fast_v(double rL, double uL, double qL, bool vacR, double uR, const CellPre &cl,
                                                      const IfaceConst &k, IfacePre &p) {
    const double um = k.um;
    p.rLc = fmax(rL, kEps);
    p.sL = cl.s; p.hL = cl.h; p.q0L = cl.q0;
    const double rsL = p.hL + p.hL;
    p.inv_rLc = rsL * rsL;
    p.ueqp_L = -um * p.hL;
    const bool vacL = rL < kEps;
    p.vacL = vacL;
    p.dU = uL - uR;
    const bool same = fabs(p.dU) < kEps;
    const bool b4 = !(vacL | vacR | same) & (uL > uR);
    p.bm = __builtin_fma(p.dU, k.inv_um, p.sL);
    const double rm = p.bm * p.bm;
    const double l0l = __builtin_fma(rL, p.ueqp_L, uL);
    const double diff = __builtin_fma(rm, uR, -(rL * uL));
    const bool zero0 = vacL | (same & !vacR);
    const bool triv = zero0 | (b4 & (diff >= 0.0)) | (!b4 & (l0l >= 0.0));
    const double bound = __builtin_fma(0.5, fabs(qL), fabs(uL) + fabs(uR)) + __builtin_fma(0.5 * um, p.sL, 0.5 * um);
    p.ci = 0; p.cfl_bad = false;
    return triv & (bound < k.cfl_lim) & k.lim_ok;
}
__device__ __forceinline__ bool arz_is_trivial_fast(double rL, double uL, double qL, double rR, double uR, const CellPre &cl,
                                                    const IfaceConst &k, IfacePre &p) {
    return arz_is_trivial_fast_v(rL, uL, qL, rR < kEps, uR, cl, k, p);
}

// flux of Q_0 and the flux Jacobian at Q_0 (darz.py:217-233; float32 entries, fp[1] = 1)
__device__ __forceinline__ void arz_flux_fp(double r0, double y0, double u0, double q0, double r0c, double h0, double inv_r0c,
                                            double um, double &Fr, double &Fy, float fp[4]) {
    Fr = r0 * u0;
    Fy = y0 * u0;
    const double ueqp_0 = -um * h0;
    const double yor = y0 * inv_r0c;
    fp[0] = (float)__builtin_fma(r0c, ueqp_0, q0);
    fp[1] = 1.f;
    fp[2] = (float)__builtin_fma(y0, ueqp_0, -(yor * yor));
    fp[3] = (float)__builtin_fma(y0 + y0, inv_r0c, q0);
}

// case 0: Q_0 = compute_Ql(Q_L) (_arz.py:155-165: u re-derived from r, y in double), dQ_0/dQ_L = I, dQ_0/dQ_R = 0
__device__ __forceinline__ void arz_trivial_fast(double rL, double yL, const IfacePre &p, const IfaceConst &k,
                                                 double &u0, double &Fr, double &Fy, float fp[4]) {
    const double qc = p.vacL ? k.ueq_2eps : p.q0L;         // u_eq(max(r_L, eps)): r_L + eps, or eps + eps under the clamp
    u0 = __builtin_fma(yL, p.inv_rLc, qc);
    arz_flux_fp(rL, yL, u0, p.q0L, p.rLc, p.hL, p.inv_rLc, k.um, Fr, Fy, fp);
}

// Q_M (case 1, compute_Qm _arz.py:184-199) and Q_C (case 2, compute_Qc :167-182) share one form: r_0 = b0^2 with
//   b0 = sqrt(r_L) + (u_L - u_R) / u_max (Q_M)   or   (u_L + u_max sqrt(r_L)) / ((gamma+1) u_max) (Q_C).
struct MidState {
    double r0, y0, u0, q0;         // Q_0
    double r0c, h0, inv_r0c;       // max(r_0, eps), 0.5 / sqrt of it, 1 / it
    double ab0, base;              // |b0| = sqrt(r_0) = r_0 ** (1 - gamma);  u_L + u_max sqrt(r_L)
};
__device__ __forceinline__ void arz_mid_state(bool c1, double uL, double uR, double sL, double bm, const IfaceConst &k, MidState &m) {
    const double um = k.um;
    m.base = __builtin_fma(um, sL, uL);
    const double b0 = c1 ? bm : m.base * k.inv_15um;
    m.ab0 = fabs(b0);
    m.r0 = b0 * b0;
    m.u0 = c1 ? uR : kGoG1 * m.base;
    m.q0 = __builtin_fma(-um, fast_sqrt(fmax(m.r0, 0.) + kEps), um);
    m.y0 = m.r0 * (m.u0 - m.q0);
    m.r0c = fmax(m.r0, kEps);
    m.h0 = (m.r0 >= kEps) ? 0.5 * fast_rcp(m.ab0) : kHalfRsqrtEps;          // 0.5 / sqrt(r0c)
    const double rs0 = m.h0 + m.h0;
    m.inv_r0c = rs0 * rs0;
}
// compute_dM (darz.py:35-122) / compute_dC (darz.py:124-192), float32 entries:
// d b0 / d(r_L, y_L) = sc * (du_L/dr_L + u_max gamma r_L^(gamma-1), du_L/dy_L), sc = 1/u_max or 1/((gamma+1) u_max);
// d y_0 = k1 d r_0 + r_0 d u_0.
__device__ __forceinline__ void arz_mid_jacobians_left(bool c1, double yL, double hL, double inv_rLc, double ueqp_L,
                                                       const MidState &m, const IfaceConst &k, float dL[4]) {
    const double um = k.um;
    const double w = __builtin_fma(-yL, inv_rLc * inv_rLc, ueqp_L) + um * hL;     // du_L/dr_L + u_max gamma r_L^(gamma-1)
    const double a2sc = (m.ab0 + m.ab0) * (c1 ? k.inv_um : k.inv_15um);
    const double k1 = __builtin_fma(m.r0, um * m.h0, m.u0 - m.q0);
    const double gL = c1 ? 0.0 : kGoG1;                // d u_0 / d u_L
    const double dr_drL = a2sc * w;
    const double dr_dyL = a2sc * inv_rLc;
    dL[0] = (float)dr_drL; dL[1] = (float)dr_dyL;
    dL[2] = (float)__builtin_fma(dr_drL, k1, m.r0 * (gL * w)); dL[3] = (float)__builtin_fma(dr_dyL, k1, m.r0 * (gL * inv_rLc));
}
// dQ_0/dQ_R, Q_M only (hR = 0.5 / sqrt(max(r_R, eps)))
__device__ __forceinline__ void arz_mid_jacobians_right(double yR, double hR, const MidState &m, const IfaceConst &k, float dR[4]) {
    const double um = k.um;
    const double a2sc = (m.ab0 + m.ab0) * k.inv_um;
    const double k1 = __builtin_fma(m.r0, um * m.h0, m.u0 - m.q0);
    const double rsR = hR + hR;
    const double inv_rRc = rsR * rsR;
    const double duR_drR = __builtin_fma(-yR, inv_rRc * inv_rRc, -um * hR);
    const double dr_drR = -a2sc * duR_drR;
    const double dr_dyR = -a2sc * inv_rRc;
    dR[0] = (float)dr_drR; dR[1] = (float)dr_dyR;
    dR[2] = (float)__builtin_fma(dr_drR, k1, m.r0 * duR_drR); dR[3] = (float)__builtin_fma(dr_dyR, k1, m.r0 * inv_rRc);
}
__device__ __forceinline__ void arz_mid_jacobians(bool c1, double yL, double yR, double hL, double inv_rLc, double ueqp_L, double hR,
                                                  const MidState &m, const IfaceConst &k, float dL[4], float dR[4]) {
    arz_mid_jacobians_left(c1, yL, hL, inv_rLc, ueqp_L, m, k, dL);
    if (c1) arz_mid_jacobians_right(yR, hR, m, k, dR);
}
// fp @ dL, fp @ dR in float32 (np.matmul); B = 0 for Q_C
__device__ __forceinline__ void arz_mid_products_left(const float fp[4], const float dL[4], float A[4]) {
    A[0] = dot2(fp[0], dL[0], fp[1], dL[2]);
    A[1] = dot2(fp[0], dL[1], fp[1], dL[3]);
    A[2] = dot2(fp[2], dL[0], fp[3], dL[2]);
    A[3] = dot2(fp[2], dL[1], fp[3], dL[3]);
}
__device__ __forceinline__ void arz_mid_products_right(const float fp[4], const float dR[4], float B[4]) {
    B[0] = dot2(fp[0], dR[0], fp[1], dR[2]);
    B[1] = dot2(fp[0], dR[1], fp[1], dR[3]);
    B[2] = dot2(fp[2], dR[0], fp[3], dR[2]);
    B[3] = dot2(fp[2], dR[1], fp[3], dR[3]);
}
__device__ __forceinline__ void arz_mid_products(bool c1, const float fp[4], const float dL[4], const float dR[4], float A[4], float B[4]) {
    arz_mid_products_left(fp, dL, A);
    if (c1) arz_mid_products_right(fp, dR, B);
    else B[0] = B[1] = B[2] = B[3] = 0.f;
}

template <bool kHavePre>
__device__ __forceinline__ void arz_interface_fast_impl(double rL, double yL, double uL, double qL,
                                                        double rR, double yR, double uR, double qR,
                                                        const CellPre *cl, const CellPre *cr,
                                                        const IfaceConst &k, Iface &o, IfaceDebug *dbg) {
    const double um = k.um;
    IfacePre p;
    if constexpr (kHavePre) arz_classify_fast(rL, uL, qL, rR, uR, *cl, k, p);
    else arz_pre_fast(rL, uL, qL, rR, uR, k, p);
    o.cfl_bad = p.cfl_bad;
    const int ci = p.ci;

    // ---- Q_0 (_arz.py:155-199, 316-326), its Jacobians (darz.py:12-192) ----
    double r0, y0, u0, q0;
    float dL[4], dR[4], fp[4];
    if (ci == 0) {
        arz_trivial_fast(rL, yL, p, k, u0, o.Fr, o.Fy, fp);
        r0 = rL; y0 = yL; q0 = p.q0L;
    } else {
        const bool c1 = (ci == 1);
        MidState m;
        arz_mid_state(c1, uL, uR, p.sL, p.bm, k, m);
        double hR = 0.;
        if (c1) {
            if constexpr (kHavePre) hR = cr->h;
            else { double sR; sqrt_hrsqrt(fmax(rR, kEps), sR, hR); }
        }
        arz_mid_jacobians(c1, yL, yR, p.hL, p.inv_rLc, p.ueqp_L, hR, m, k, dL, dR);
        arz_flux_fp(m.r0, m.y0, m.u0, m.q0, m.r0c, m.h0, m.inv_r0c, um, o.Fr, o.Fy, fp);
        r0 = m.r0; y0 = m.y0; u0 = m.u0; q0 = m.q0;
    }

    if (dbg) {
        dbg->ci = ci;
        dbg->q0[0] = r0; dbg->q0[1] = y0; dbg->q0[2] = u0; dbg->q0[3] = q0;
        arz_speeds_ref(rL, uL, qL, rR, uR, um, dbg->speed[0], dbg->speed[1]);
        for (int j = 0; j < 4; ++j) {
            dbg->fp[j] = fp[j];
            dbg->dL[j] = (ci == 0) ? ((j == 0 || j == 3) ? 1.f : 0.f) : dL[j];
            dbg->dR[j] = (ci == 1) ? dR[j] : 0.f;
        }
    }
    if (ci == 0) {          // dL = I, dR = 0: the products are fp and 0 exactly
        o.A[0] = fp[0]; o.A[1] = fp[1]; o.A[2] = fp[2]; o.A[3] = fp[3];
        o.B[0] = o.B[1] = o.B[2] = o.B[3] = 0.f;
    } else {
        arz_mid_products(ci == 1, fp, dL, dR, o.A, o.B);
    }
}

__device__ __forceinline__ void arz_interface_fast(double rL, double yL, double uL, double qL,
                                                   double rR, double yR, double uR, double qR,
                                                   const IfaceConst &k, Iface &o, IfaceDebug *dbg = nullptr) {
    arz_interface_fast_impl<false>(rL, yL, uL, qL, rR, yR, uR, qR, nullptr, nullptr, k, o, dbg);
}
// the same with the cells' density-only quantities supplied (bitwise the same results: arz_cell_pre is what the other form calls)
__device__ __forceinline__ void arz_interface_fast_pre(double rL, double yL, double uL, double qL, const CellPre &cl,
                                                       double rR, double yR, double uR, double qR, const CellPre &cr,
                                                       const IfaceConst &k, Iface &o) {
    arz_interface_fast_impl<true>(rL, yL, uL, qL, rR, yR, uR, qR, &cl, &cr, k, o, nullptr);
}

// arz_interface_fast_pre for a wavefront of QUEUED interfaces (phase 2 of the pair rollout kernel): the same operations on the
// same operands -- results are bit-identical -- laid out for a short instruction stream of ONE wavefront (phase 2 is a serial
// section: its length in instructions is what a step pays for it):
//   * the case logic runs on lane masks (64-bit scalar values from ballots), never through selects of booleans in vector registers;
//   * Q_M / Q_C with their Jacobians are evaluated by every lane without branches (the few lanes whose interface turns out
//     trivial -- the interfaces a queue always holds -- are overwritten at the end, in a block a wavefront without such a lane skips);
//   * the cells' float32 state is taken as stored (no float64 copies of values only compared in float32).
// ls / rs = (r, y, u, u_eq) of the left / right cell, (sL, hL, q0L) = the left cell's CellPre, hR = 0.5 / sqrt(max(r_R, eps)).
__device__ __forceinline__ void arz_interface_queued(const float4 ls, double sL, double hL, double q0L, const float4 rs, double hR,
                                                     const IfaceConst &k, Iface &o) {
    typedef unsigned long long M;
    const double um = k.um;
    const double rL = (double)ls.x, yL = (double)ls.y, uL = (double)ls.z, qL = (double)ls.w;
    const double yR = (double)rs.y, uR = (double)rs.z;
    const double rLc = fmax(rL, kEps);
    const double rsL = hL + hL;
    const double inv_rLc = rsL * rsL;
    const double ueqp_L = -um * hL;
    // ---- case logic (arz_classify_fast) on lane masks ----
    const M vacL = __builtin_amdgcn_ballot_w64(ls.x <= kEpsF);          // r < 1e-5 (double) <=> r <= float32(1e-5) for a float32 r
    const M vacR = __builtin_amdgcn_ballot_w64(rs.x <= kEpsF);
    const double dU = uL - uR;
    const M same = __builtin_amdgcn_ballot_w64(fabs(dU) < kEps);
    const M wave_m = ~(vacL | vacR | same);
    const M b4 = wave_m & __builtin_amdgcn_ballot_w64(ls.z > rs.z);
    const double qm_u = um + uL - qL;
    const M b5 = wave_m & ~b4 & __builtin_amdgcn_ballot_w64(qm_u > uR);
    const double bm = __builtin_fma(dU, k.inv_um, sL);
    const double rm = bm * bm;
    const double abm = fabs(bm);
    const double l0l = __builtin_fma(rL, ueqp_L, uL);
    const double diff = __builtin_fma(rm, uR, -(rL * uL));
    const double l0m = (rm >= kEps) ? __builtin_fma(-0.5 * um, abm, uR) : __builtin_fma(rm, -um * kHalfRsqrtEps, uR);
    const M l0l_ok = __builtin_amdgcn_ballot_w64(l0l >= 0.0);
    const M zero0 = vacL | (same & ~vacR);
    const M trivM = zero0 | (b4 & __builtin_amdgcn_ballot_w64(diff >= 0.0)) | (~b4 & l0l_ok);
    const M midM = b4 | (b5 & __builtin_amdgcn_ballot_w64(l0m <= 0.0));
    const bool triv = __builtin_amdgcn_inverse_ballot_w64(trivM);
    const bool c1 = __builtin_amdgcn_inverse_ballot_w64(midM);
    // ---- CFL (arz_classify_fast): the exact tests only where the cheap bound does not settle it ----
    const double bound = __builtin_fma(0.5, fabs(qL), fabs(uL) + fabs(uR)) + __builtin_fma(0.5 * um, sL, 0.5 * um);
    const bool sure = (bound < k.cfl_lim) & k.lim_ok;
    o.cfl_bad = false;
    if (__builtin_amdgcn_ballot_w64(!sure)) {
        asm volatile("" ::: "memory");
        const bool vL = __builtin_amdgcn_inverse_ballot_w64(vacL), vR = __builtin_amdgcn_inverse_ballot_w64(vacR);
        const bool is4 = __builtin_amdgcn_inverse_ballot_w64(b4), is5 = __builtin_amdgcn_inverse_ballot_w64(b5);
        const bool z0 = __builtin_amdgcn_inverse_ballot_w64(zero0);
        const double den = fmax(rm - rL, kEps);
        const bool ok_b4 = fabs(diff) < k.cfl_lim * den;
        const bool ok_m = fabs(l0l + l0m) < k.cfl_lim2;
        const bool ok_q = fabs(l0l + qm_u) < k.cfl_lim2;
        const bool ok_avg = (is5 & ok_m) | (!is5 & ok_q);
        const bool ok0 = z0 | (is4 & ok_b4) | (!is4 & ok_avg);
        const bool ok_uL = fabs(uL) < k.cfl_lim, ok_uR = fabs(uR) < k.cfl_lim;
        const bool ok1 = (vL & ok_uL) | (!vL & ((vR & ok_avg) | (!vR & ok_uR)));
        o.cfl_bad = !(ok0 & ok1 & k.lim_ok);
    }
    // ---- Q_M (c1) / Q_C, their Jacobians, the flux and its Jacobian, the products: every lane (arz_mid_state ... arz_mid_products) ----
    MidState m;
    arz_mid_state(c1, uL, uR, sL, bm, k, m);
    float dL[4], dR[4], fp[4];
    arz_mid_jacobians_left(c1, yL, hL, inv_rLc, ueqp_L, m, k, dL);
    arz_mid_jacobians_right(yR, hR, m, k, dR);
    arz_flux_fp(m.r0, m.y0, m.u0, m.q0, m.r0c, m.h0, m.inv_r0c, um, o.Fr, o.Fy, fp);
    arz_mid_products_left(fp, dL, o.A);
    float B[4];
    arz_mid_products_right(fp, dR, B);
    o.B[0] = c1 ? B[0] : 0.f; o.B[1] = c1 ? B[1] : 0.f; o.B[2] = c1 ? B[2] : 0.f; o.B[3] = c1 ? B[3] : 0.f;
    // ---- Q_0 = Q_L where the interface is trivial after all ----
    if (__builtin_amdgcn_ballot_w64(triv)) {
        asm volatile("" ::: "memory");
        if (triv) {
            const double qc = __builtin_amdgcn_inverse_ballot_w64(vacL) ? k.ueq_2eps : q0L;
            const double u0 = __builtin_fma(yL, inv_rLc, qc);
            float fpt[4];
            arz_flux_fp(rL, yL, u0, q0L, rLc, hL, inv_rLc, um, o.Fr, o.Fy, fpt);
            o.A[0] = fpt[0]; o.A[1] = fpt[1]; o.A[2] = fpt[2]; o.A[3] = fpt[3];
            o.B[0] = o.B[1] = o.B[2] = o.B[3] = 0.f;
        }
    }
}

__device__ __forceinline__ void arz_interface(double rL, double yL, double uL, double qL,
                                              double rR, double yR, double uR, double qR,
                                              const IfaceConst &k, Iface &o, IfaceDebug *dbg = nullptr) {
#ifdef DHTS_IEEE_DIV_SQRT
    arz_interface_ieee(rL, yL, uL, qL, rR, yR, uR, qR, k, o, dbg);
#else
    arz_interface_fast(rL, yL, uL, qL, rR, yR, uR, qR, k, o, dbg);
#endif
}

}  // namespace dhts
