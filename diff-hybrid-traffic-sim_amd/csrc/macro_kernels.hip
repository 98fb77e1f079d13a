// macro_kernels.hip -- time-fused forward and reverse sweeps of the ARZ cell stencil on gfx950.
//
// Forward (macro_rollout_fwd_kernel): one workgroup of W wavefronts owns one traffic lane for the whole rollout.
//   The lane's state (r, y, u, u_eq; float32, N cells + 2 ghosts) lives in LDS (ping-pong) for all T steps; HBM
//   sees only the initial load, the final store and the Jacobian tape (48 B per cell-step, written as three
//   coalesced 16-B-per-lane streams).  Each wave owns a contiguous chunk of cells and walks it in 64-interface
//   passes: thread t of a pass solves ONE interface (between cells i-1 and i) once, hands the result to thread
//   t-1 with a DPP wave shift (the cell left of an interface needs it as ITS right interface), and thread 63
//   takes the first interface of the previous pass from scalar registers.  One workgroup barrier per time
//   step; no inter-workgroup traffic.
// Reverse (macro_rollout_bwd_kernel): one workgroup per lane, cotangent (g_r, g_y) in LDS, tape read back as the
//   same three streams, newest step first; g' = J^T g with the reference's float32 accumulation order.
//
// Reference: road/lane/_macro_lane.py:83-146, road/lane/dmacro_lane.py:96-132 and :277-309.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/dhts.h"
#include "arz_device.hpp"

namespace dhts {

// wave_shl:1 DPP move: lane t receives lane t+1's value; lane 63 (no source) keeps `carry`.
// One v_mov_b32_dpp per dword -- no LDS round trip.
constexpr int kDppWaveShl1 = 0x130;
__device__ __forceinline__ float take_right(float x, float carry) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(x), kDppWaveShl1, 0xF, 0xF, false));
}
__device__ __forceinline__ double take_right(double x, double carry) {
    const long long xb = __double_as_longlong(x), cb = __double_as_longlong(carry);
    const int lo = __builtin_amdgcn_update_dpp((int)(cb & 0xffffffffll), (int)(xb & 0xffffffffll), kDppWaveShl1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(cb >> 32), (int)(xb >> 32), kDppWaveShl1, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float bcast0(float x) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ double bcast0(double x) {
    long long b = __double_as_longlong(x);
    int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
    int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ void raise_fault(dhts_error *err, int code, int step, int lane, int index) {
    if (err == nullptr) return;
    if (atomicCAS(&err->code, 0, code) == 0) {
        err->step = step;
        err->lane = lane;
        err->index = index;
    }
}

// ---- the rollout tape -----------------------------------------------------------------------------------------------
// What the reverse sweep needs of a step is, per interface i, A_i = flux'(Q_0) dQ_0/dQ_L and B_i = flux'(Q_0) dQ_0/dQ_R
// (dmacro_lane.py:116-124).  For a trivial interface (Q_0 = Q_L; 85-89 % of them in BASELINE config 2) dQ_0/dQ_L = I and
// dQ_0/dQ_R = 0: A_i is the flux Jacobian at Q_L, whose entry [0][1] is the constant 1 (darz.py:217-233), and B_i = 0 --
// three floats instead of eight.  The other interfaces ("exceptions") keep their (A, B), compacted in the order the forward
// solved them.  One row per (step, lane), every block on a 128-byte line:
//   S  float [N][3]                                fp[0], fp[2], fp[3] of interface i < N where it is trivial
//   H  u32 cnt, u32 0 | u16 idx [N + 1]            idx[j] = interface of exception j < cnt; interface N is always one
//   E  float4 [N + 1][2]                           (A, B) of exception j; only the first cnt entries are written / read
// 8.8 KB of traffic per 512-cell row against the 16.6 KB of a dense (A, B) tape and the 24.6 KB of the reference's blocks.
struct TapeGeom {
    int s_f4, h_f4, e_f4;
    size_t row_f4;
};
__host__ __device__ inline TapeGeom tape_geom(int N) {
    TapeGeom g;
    g.s_f4 = (((3 * N + 3) >> 2) + 7) & ~7;
    g.h_f4 = ((8 + 2 * (N + 1) + 127) >> 7) << 3;
    g.e_f4 = (2 * (N + 1) + 7) & ~7;
    g.row_f4 = (size_t)g.s_f4 + g.h_f4 + g.e_f4;
    return g;
}
__device__ __forceinline__ unsigned *tape_hdr(float4 *row, const TapeGeom &g) { return reinterpret_cast<unsigned *>(row + g.s_f4); }
__device__ __forceinline__ const unsigned *tape_hdr(const float4 *row, const TapeGeom &g) { return reinterpret_cast<const unsigned *>(row + g.s_f4); }
__device__ __forceinline__ unsigned short *tape_idx(unsigned *hdr, const TapeGeom &g) { return reinterpret_cast<unsigned short *>(hdr + 2); }
__device__ __forceinline__ const unsigned short *tape_idx(const unsigned *hdr, const TapeGeom &g) { return reinterpret_cast<const unsigned short *>(hdr + 2); }
struct __attribute__((packed, aligned(4))) TapeFp { float f0, f2, f3; };
static_assert(sizeof(TapeFp) == 12, "TapeFp layout");
// (stored member by member -- three dword stores: measured faster in the forward kernel than one global_store_dwordx3 and
// than three planes, 4.32 against 4.41 and 4.44 ms)
__device__ __forceinline__ float4 tape_trivial_A(const TapeFp &s) { return make_float4(s.f0, 1.f, s.f2, s.f3); }

// grid = L workgroups (one traffic lane each) of W = blockDim.x / 64 wavefronts; dynamic LDS = 2 * 4 * (N + 2) floats.
// Wave w owns the cells [w C, min(N, (w + 1) C)) with C = 64 p - 1, i.e. at most 64 p interfaces = p passes, so no
// wave ever needs a pass for a single left-over interface; the state is ping-pong buffered in LDS and the only
// synchronisation is one workgroup barrier per time step.
template <bool kIface>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(5, 5))) void macro_rollout_fwd_kernel(
    int L, int N, int T, int p, double dt, double dx, double um,
    const float *__restrict__ r_in, const float *__restrict__ y_in, const float *__restrict__ u_in,
    const float *__restrict__ q_in, const float *__restrict__ ghost,
    float *__restrict__ r_out, float *__restrict__ y_out, float *__restrict__ u_out, float *__restrict__ q_out,
    float4 *__restrict__ tape, float *__restrict__ hist, dhts_error *err) {
    extern __shared__ float lds[];
    const int lane = blockIdx.x;
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int wv = tid >> 6;
    const int P = N + 2;
    const size_t base = (size_t)lane * N;

    // buffer b, plane k (r, y, u, u_eq) at lds + (b * 4 + k) * P; index c + 1 holds cell c, 0 and N + 1 the ghosts
    for (int k = tid; k < N; k += blockDim.x) {
        lds[0 * P + k + 1] = r_in[base + k];
        lds[1 * P + k + 1] = y_in[base + k];
        lds[2 * P + k + 1] = u_in[base + k];
        lds[3 * P + k + 1] = q_in[base + k];
    }
    if (tid < 16) {
        const int b = tid >> 3, side = (tid >> 2) & 1, k = tid & 3;
        lds[(b * 4 + k) * P + (side ? N + 1 : 0)] = ghost[(size_t)lane * 8 + side * 4 + k];
    }
    __syncthreads();

    const int C = 64 * p - 1;
    const int lo = wv * C;                           // first cell / first interface of this wave
    const int hi = (lo + C < N) ? lo + C : N;        // one past its last cell = its last interface
    const int K = (lo < N) ? ((hi - lo + 1 + 63) >> 6) : 0;
    const int Np = (N + 63) & ~63;
    const TapeGeom geo = tape_geom(N);
    const size_t tape_row = kIface ? geo.row_f4 : (size_t)3 * Np;
    const double c = dt / dx;                        // update_coefficient, _macro_lane.py:99
    const float cf = (float)c, ncf = (float)(-c);
    const float umf = (float)um;
    IfaceConst kc;
    kc.set_um(um); kc.set_grid(dt, dx);
    int fault_step = -1, fault_index = 0;
    // an upstream ghost in double (ghost_source_pack, arz_device.hpp: the itscp source lanes of the stepwise network paths)
    const bool src_ghost = ghost_is_source(ghost[(size_t)lane * 8]);
    double src_r = 0., src_y = 0., src_u = 0., src_q = 0.;
    if (src_ghost) ghost_source_unpack(ghost[(size_t)lane * 8 + 2], ghost[(size_t)lane * 8 + 3], um, src_r, src_y, src_u, src_q);

    for (int step = 0; step < T; ++step) {
        const float *cur = lds + (step & 1) * 4 * P;
        float *nxt = lds + ((step & 1) ^ 1) * 4 * P;
        const float *Sr = cur, *Sy = cur + P, *Su = cur + 2 * P, *Sq = cur + 3 * P;
        // first interface of the pass above (j + 1), broadcast from its thread 0
        double cFr = 0., cFy = 0.;
        float cA0 = 0.f, cA1 = 0.f, cA2 = 0.f, cA3 = 0.f, cB0 = 0.f, cB1 = 0.f, cB2 = 0.f, cB3 = 0.f;
        float4 *tp = tape ? tape + ((size_t)step * L + lane) * tape_row : nullptr;
        float *hp = hist ? hist + ((size_t)step * L + lane) * 3 * N : nullptr;
        if (kIface && tp) {
            // this kernel keeps every interface as an exception: interface N at slot 0, interface i < N at slot i + 1
            unsigned *H = tape_hdr(tp, geo);
            if (tid == 0) { H[0] = (unsigned)(N + 1); H[1] = 0u; }
        }
        if (K > 0) {
            for (int j = K - 1; j >= 0; --j) {
                const int i = lo + (j << 6) + t;   // interface i, and cell i to its right
                const bool vi = i <= hi;
                const bool vc = i < hi;
                const int ip = vi ? i : hi;
                double rL = Sr[ip], yL = Sy[ip], uL = Su[ip], qL = Sq[ip];
                if (src_ghost && ip == 0) { rL = src_r; yL = src_y; uL = src_u; qL = src_q; }
                const double rR = Sr[ip + 1], yR = Sy[ip + 1], uR = Su[ip + 1], qR = Sq[ip + 1];
                Iface f;
                arz_interface(rL, yL, uL, qL, rR, yR, uR, qR, kc, f);
                // CFL: dt < dx / max(|speed|, 1e-5) for both speeds (_macro_lane.py:141-146)
                if (vi && fault_step < 0 && f.cfl_bad) { fault_step = step; fault_index = i; }

                // right interface of cell i = interface i + 1: thread t + 1, or the carried one for thread 63
                const double Fr_R = take_right(f.Fr, cFr), Fy_R = take_right(f.Fy, cFy);
                cFr = bcast0(f.Fr); cFy = bcast0(f.Fy);
                float A0 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f, B0 = 0.f, B1 = 0.f, B2 = 0.f, B3 = 0.f;
                if constexpr (!kIface) {
                    A0 = take_right(f.A[0], cA0); A1 = take_right(f.A[1], cA1);
                    A2 = take_right(f.A[2], cA2); A3 = take_right(f.A[3], cA3);
                    B0 = take_right(f.B[0], cB0); B1 = take_right(f.B[1], cB1);
                    B2 = take_right(f.B[2], cB2); B3 = take_right(f.B[3], cB3);
                    cA0 = bcast0(f.A[0]); cA1 = bcast0(f.A[1]); cA2 = bcast0(f.A[2]); cA3 = bcast0(f.A[3]);
                    cB0 = bcast0(f.B[0]); cB1 = bcast0(f.B[1]); cB2 = bcast0(f.B[2]); cB3 = bcast0(f.B[3]);
                } else if (tp && vi) {
                    // the two 2x2 products of interface i; the reverse sweep forms the cell blocks from them
                    const int slot = (i == N) ? 0 : i + 1;
                    float4 *E = tp + geo.s_f4 + geo.h_f4;
                    E[2 * slot] = make_float4(f.A[0], f.A[1], f.A[2], f.A[3]);
                    E[2 * slot + 1] = make_float4(f.B[0], f.B[1], f.B[2], f.B[3]);
                    tape_idx(tape_hdr(tp, geo), geo)[slot] = (unsigned short)i;
                }

                if (vc) {
                    // Godunov update, _macro_lane.py:109-112, float32 store :327-334
                    const float nr = (float)(rR + (f.Fr - Fr_R) * c);
                    const float ny = (float)(yR + (f.Fy - Fy_R) * c);
                    float nu, nq;
                    glue_from_r_y(nr, ny, umf, nu, nq);      // set_next_state_vector_y, :282-299
                    nxt[i + 1] = nr; nxt[P + i + 1] = ny; nxt[2 * P + i + 1] = nu; nxt[3 * P + i + 1] = nq;
                    if (!kIface && tp) {
                        // dMacroLane._backward, dmacro_lane.py:126-129
                        float4 d0, d1, d2;
                        d0.x = ncf * (-f.A[0]); d0.y = ncf * (-f.A[1]); d0.z = ncf * (-f.A[2]); d0.w = ncf * (-f.A[3]);
                        d2.x = ncf * B0; d2.y = ncf * B1; d2.z = ncf * B2; d2.w = ncf * B3;
                        d1.x = 1.f - cf * (A0 - f.B[0]); d1.y = 0.f - cf * (A1 - f.B[1]);
                        d1.z = 0.f - cf * (A2 - f.B[2]); d1.w = 1.f - cf * (A3 - f.B[3]);
                        tp[i] = d0;
                        tp[Np + i] = d1;
                        tp[2 * Np + i] = d2;
                    }
                    if (hp) { hp[i] = nr; hp[N + i] = ny; hp[2 * N + i] = nu; }
                }
            }
        }
        __syncthreads();
    }
    const float *fin = lds + (T & 1) * 4 * P;
    for (int k = tid; k < N; k += blockDim.x) {
        r_out[base + k] = fin[k + 1];
        y_out[base + k] = fin[P + k + 1];
        u_out[base + k] = fin[2 * P + k + 1];
        q_out[base + k] = fin[3 * P + k + 1];
    }
    if (fault_step >= 0) raise_fault(err, DHTS_FAULT_CFL, fault_step, lane, fault_index);
}

// ---- the rollout forward kernel: cheap solve everywhere, full solve only where needed ---------------------------------
__device__ __forceinline__ void lds_only_barrier() {     // global stores (the tape) stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Orders this wavefront's LDS stores in front of its later LDS loads FOR THE COMPILER: lanes of one wavefront hand data to
// each other through LDS (a cell's record to its right neighbour's thread, a pass's records and fluxes to the next pass), and
// within one thread those addresses never alias, so nothing else stops the scheduler from hoisting the loads.  The hardware
// executes a wavefront's LDS operations in order; no instruction is emitted.
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One cell's record in LDS (48 bytes; a 48-byte stride keeps 16-byte accesses of 16 consecutive lanes on distinct banks):
// the float32 state and what the interface to the cell's right needs of it, computed once by the cell's owner.
struct __attribute__((aligned(16))) CellRec {
    float4 st;       // r, y, u, u_eq  (FullQ, _arz.py:10-115)
    double2 sh;      // CellPre s, h
    double2 q0;      // CellPre q0 (+ padding: every access is a conflict-free 16-byte one)
};
static_assert(sizeof(CellRec) == 48, "CellRec layout");

// New (u, u_eq) of a cell from its new (r, y) -- FullQ.set_r_y, _arz.py:88-92, float32 torch arithmetic -- together with
// the cell's CellPre.  The float32 square root and division of the glue come out of the double-precision square roots
// CellPre needs anyway: sqrtf(t32) = fl32(sE + (t32 - tE) hE) with sE = sqrt(tE), hE = 0.5 / sE, tE = r + eps in double
// (first-order in t32 - tE = O(1e-7): the remainder is 5e-16 relative), y / r = fl32(y (2 h)^2).  Both agree with the
// correctly rounded float32 result unless it lies within 2e-14 (relative) of a rounding boundary, i.e. except with
// probability ~1e-6 per operation -- the level of the fast double layer (fast_math.hpp); the IEEE forms are glue_from_r_y.
__device__ __forceinline__ void cell_glue_pre(float r, float y, float umf, const IfaceConst &k, float &u, float &q, CellPre &c) {
    const double rd = (double)r;
    sqrt_hrsqrt(fmax(rd, kEps), c.s, c.h);
    const double tE = fmax(rd, 0.) + kEps;
    double sE, hE;
    sqrt_hrsqrt(tE, sE, hE);
    c.q0 = __builtin_fma(-k.um, sE, k.um);
    const float t32 = r + kEpsF;
    const float s32 = (float)__builtin_fma((double)t32 - tE, hE, sE);
    const float one_m = 1.f - s32;
    q = (0.f > r) ? k.ueq_neg_f : umf * one_m;              // max(r, 0.) picked the Python float 0. (glue_u_eq)
    const double rs = c.h + c.h;
    const double inv = (r <= kEpsF) ? k.inv_epsf : rs * rs;  // 1 / max(r, eps32): r below OR AT float32(eps) divides by float32(eps)
    const float yor = (float)((double)y * inv);
    u = yor + ((r < kEpsF) ? k.ueq_2eps_f : q);
}

// grid = L workgroups (one traffic lane each) of W = blockDim.x / 64 wavefronts; wave w owns the cells [64 p w, 64 p (w + 1)),
// thread t of its pass j the cell c = 64 p w + 64 j + t and that cell's LEFT interface (interface c, between cells c-1 and c).
// In BASELINE config 2, 89 % of the interface solves end in the trivial case (Q_0 = Q_L, dQ_0/dQ_L = I, dQ_0/dQ_R = 0), but
// a 64-wide pass is uniformly trivial only 23 % of the time: evaluated in place, the Q_M / Q_C states with their Jacobians
// (more than half of the work) are paid for by every lane.  And on this chip every vector instruction of a wavefront costs
// about 4 cycles whatever its type (PMC: profiles/r02*), so the instruction COUNT is what a step costs, and the chain
// phase 1 -> phase 2 -> phase 1 ... of ONE lane is what bounds it when few lanes share a CU.  A step therefore runs as
//   phase 1 (every cell's thread): finish the previous step -- new (r, y) from the two fluxes of the cell, (u, u_eq) and
//     the cell's CellPre from one pair of square roots, written to the cell's record -- then read the left neighbour's
//     record, decide whether the interface is trivial, and if so write its flux (LDS) -- its tape entry is the three
//     non-constant entries of the flux Jacobian; otherwise append it to the lane's queue;
//   phase 2 (the first threads of the workgroup, one queue entry each, at raised priority: they are the workgroup's
//     critical path): the full solve for the queued interfaces, their fluxes to LDS, their tape entries to HBM.  Always
//     queued: the first interface of every wave's chunk (its left cell belongs to another wave) and interface N (no cell
//     owns it).
// One queue pass serves the whole lane (8 passes of phase 1 at 512 cells), so the expensive path runs about once per 64
// non-trivial interfaces instead of once per 64 interfaces.  Two LDS-only barriers per step.
// Where the time went when a wavefront made one pass (s_memtime stamps, config 2, 8 waves per lane, 2 workgroups per CU,
// 4.4 ms): phase 1 ~2000 cycles (every SIMD busy: instruction throughput), phase 2 ~1800 cycles on the one or two waves that
// hold queue entries while the others wait -- a dependent chain (three LDS round trips, the classification, two rsq / rcp
// chains, the Jacobians), latency not instruction count -- and ~600 cycles of barrier skew.  Since then (3.4 ms): the scalar
// diet (running pointers, literal flags: every instruction of a wavefront, scalar ones included, takes a turn of its
// issue slot) and two unrolled passes per wavefront (two independent cells per thread to interleave; four-wave workgroups,
// all lanes of config 2 resident at once).  Measured and dropped (tools/probes/*.hip.txt; times against the kernel of their day):
//   * fluxes only in phase 2, Jacobians one step behind on another wave: the Jacobian chain is as long as the full solve,
//     so the phase does not get shorter; a fifth / ninth dedicated wavefront per workgroup costs a workgroup of occupancy;
//   * the solve cut into three roles (flux | A | B) on three waves side by side: every role still takes ~1700 cycles;
//   * 6 or 8 waves per SIMD (80 / 64 VGPRs): no gain / spills;
//   * dedicated solver wavefronts (4 cell waves x 2 passes + 4 solver waves; the flux in front of the barrier that releases
//     the next phase 1, the Jacobians of the same entries behind it): 4.7 ms against 4.4 -- with half the waves owning
//     cells, phase 1 hides its own latency worse than phase 2 gets shorter; 8 + 4 waves need <= 80 VGPRs for two
//     workgroups per CU, which spills 35 registers;
//   * per-wave queues, every wave solving what it queued itself (no serial section, all waves busy in phase 2): 5.2 ms with
//     2 waves per lane, 5.9 with 8 -- the full solve's instruction stream is then paid once per wave instead of once per lane;
//   * the trivial interfaces' tape entries (three floats of the flux Jacobian) formed in phase 2 by the wavefronts that hold no
//     queue entry instead of in phase 1: 4.48 ms against 4.32 -- with two workgroups per CU those wavefronts are not idle
//     time, they are the other workgroup's phase 1;
//   * (round 3; all bitwise equal to the kernel below, gpurun_out/r03b_variants.log) the queue dealt round-robin over all four
//     wavefronts, 19 entries each, instead of 64 to the first and the rest to the second: 3.84 ms against 3.46 -- a wave
//     instruction costs its issue slot whatever its EXEC mask, so four wavefronts pay for the solve's stream instead of two;
//     phase 1 staged over both passes (every update, then every interface, then ONE queue append per wavefront): 3.50 --
//     the passes' streams interleave but the append was not what the wavefront waited for;
//   * two traffic lanes per workgroup half a step apart (phase 1 of one lane beside phase 2 of the other in every barrier
//     interval, 4 + 4 wavefronts, two phase-1 passes per thread interleaved stage by stage): 6.1 ms -- an interval takes
//     ~5100 cycles for the phase-1 waves and ~4000-4700 for the phase-2 waves that share their SIMDs, against 2000 + 1800
//     when the phases alternate: neither chain got shorter by running beside the other.
// Dynamic LDS: cell records [N + 2] (index c + 1 = cell c, 0 and N + 1 the ghosts; updated in place) | flux double2 [N + 1]
// | queue int [N + 2] | 2 counters.
// kP: 64-cell passes per wavefront as a literal (1 or 2; 0 = the run-time value p_arg): the pass loop is unrolled, the
// per-thread addresses are loop invariants, and with two passes a wavefront has two independent cells' instruction streams
// to interleave; kFull: every thread-pass owns a cell (N = kP blockDim.x): no validity masks; kHist: the state history may
// be asked for.  The step body is instantiated with literal (finish the previous step, start this one) flags: first step,
// steps in between, last.
template <int kP, bool kFull, bool kHist>
__global__ __launch_bounds__(1024) void macro_rollout_fwd2_kernel(
    int L, int N, int T, int p_arg, double dt, double dx, double um,
    const float *__restrict__ r_in, const float *__restrict__ y_in, const float *__restrict__ u_in,
    const float *__restrict__ q_in, const float *__restrict__ ghost,
    float *__restrict__ r_out, float *__restrict__ y_out, float *__restrict__ u_out, float *__restrict__ q_out,
    float4 *__restrict__ tape, float *__restrict__ hist, dhts_error *err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = blockIdx.x;
    const int tid = threadIdx.x;
    const int t = tid & 63;
    const int wv = tid >> 6;
    const int Wc = blockDim.x >> 6;                  // wavefronts
    const int ncell = blockDim.x;
    const size_t base = (size_t)lane * N;
    CellRec *CR = reinterpret_cast<CellRec *>(smem);
    double2 *FX = reinterpret_cast<double2 *>(CR + (N + 2));
    int *Q = reinterpret_cast<int *>(FX + (N + 1));
    int *CNT = Q + (N + 2);                          // queue length by step parity
    IfaceConst kc;
    kc.set_um(um); kc.set_grid(dt, dx);

    for (int k = tid; k < N + 2; k += blockDim.x) {
        float4 st;
        if (k == 0 || k == N + 1) {
            const float *g = ghost + (size_t)lane * 8 + (k ? 4 : 0);
            st = make_float4(g[0], g[1], g[2], g[3]);
        } else {
            st = make_float4(r_in[base + k - 1], y_in[base + k - 1], u_in[base + k - 1], q_in[base + k - 1]);
        }
        CellPre c;
        arz_cell_pre((double)st.x, um, c);
        CR[k].st = st;
        CR[k].sh = make_double2(c.s, c.h);
        CR[k].q0 = make_double2(c.q0, 0.);
    }
    if (tid == 0) { Q[0] = N; CNT[0] = 1; CNT[1] = 1; }      // entry 0 of every step's queue: interface N
    __syncthreads();

    const TapeGeom geo = tape_geom(N);
    const int p = kP > 0 ? kP : p_arg;

    const int lo = wv * (p << 6);                    // first cell of this wave
    const double c = dt / dx;                        // update_coefficient, _macro_lane.py:99
    const float umf = (float)um;
    int fault_step = -1, fault_index = 0;
    int rot = 0;                                     // the cell wave that takes the head of the queue rotates with the step

    // the tape row / history block of the step, advanced by one step's worth per trip (no 64-bit multiplications in the loop)
    float4 *tp_run = tape ? tape + (size_t)lane * geo.row_f4 : nullptr;
    float *hp_run = (kHist && hist) ? hist + (size_t)lane * 3 * N : nullptr;
    const size_t tp_stride = (size_t)L * geo.row_f4, hp_stride = (size_t)L * 3 * N;

    // (r, y) of a thread's own cells as doubles, carried from step to step: a thread owns the same cells for the whole rollout,
    // so the update neither re-reads them from the record nor widens them again (literal pass counts only; 3.46 -> 3.39 ms)
    constexpr bool kKeep = kP > 0;
    double rd_own[kP > 0 ? kP : 1], yd_own[kP > 0 ? kP : 1];

    auto body = [&](auto upd_c, auto solve_c, const int n) {
        constexpr bool upd = decltype(upd_c)::value;       // finish step n - 1
        constexpr bool solve = decltype(solve_c)::value;   // start step n
        float4 *tp = solve ? tp_run : nullptr;
        TapeFp *tS = reinterpret_cast<TapeFp *>(tp);
        unsigned *tH = tape_hdr(tp, geo);
        float *hp = (kHist && upd) ? hp_run : nullptr;
        if (tp_run) tp_run += tp_stride;
        if (kHist && hp_run && upd) hp_run += hp_stride;
        int *cnt = CNT + (n & 1);
#pragma unroll
        for (int j = 0; j < p; ++j) {
            const int i = lo + (j << 6) + t;                // cell i and its left interface i
            const bool vc = kFull || i < N;
            const unsigned ic = (unsigned)(vc ? i : N - 1);
            CellRec *own = CR + ic + 1;
            float4 st;
            if (!(kKeep && upd)) st = own->st;
            if (kKeep && !upd) { rd_own[kP > 0 ? j : 0] = (double)st.x; yd_own[kP > 0 ? j : 0] = (double)st.y; }
            if (upd) {
                // Godunov update, _macro_lane.py:109-112, float32 store :327-334
                const double2 Fl = FX[ic], Fr = FX[ic + 1];
                const double r_old = kKeep ? rd_own[kP > 0 ? j : 0] : (double)st.x;
                const double y_old = kKeep ? yd_own[kP > 0 ? j : 0] : (double)st.y;
                st.x = (float)(r_old + (Fl.x - Fr.x) * c);
                st.y = (float)(y_old + (Fl.y - Fr.y) * c);
                if (kKeep) { rd_own[kP > 0 ? j : 0] = (double)st.x; yd_own[kP > 0 ? j : 0] = (double)st.y; }
                CellPre cp;
                cell_glue_pre(st.x, st.y, umf, kc, st.z, st.w, cp);     // set_next_state_vector_y, :282-299
                if (vc && solve) { own->st = st; own->sh = make_double2(cp.s, cp.h); own->q0 = make_double2(cp.q0, 0.); }
                if (vc && hp) { hp[i] = st.x; hp[N + i] = st.y; hp[2 * N + i] = st.z; }
            }
            if (!solve) {
                if (vc) { r_out[base + i] = st.x; y_out[base + i] = st.y; u_out[base + i] = st.z; q_out[base + i] = st.w; }
                continue;
            }
            // the left neighbour at this time level: written just above by the lane below, or in the previous pass (a wave's
            // LDS operations complete in order); the first cell of the chunk has its left neighbour in another wave: queued
            wave_lds_handoff();
            const CellRec *lf = CR + ic;
            const float4 ls = lf->st;
            const double2 lsh = lf->sh, lq0 = lf->q0;
            CellPre cl;
            cl.s = lsh.x; cl.h = lsh.y; cl.q0 = lq0.x;
            IfacePre pre;
            const bool easy = arz_is_trivial_fast((double)ls.x, (double)ls.z, (double)ls.w, (double)st.x, (double)st.z, cl, kc, pre);
            const bool triv = vc & easy & !((j == 0) & (t == 0));
            // the trivial solve runs on every lane (the few that are not trivial sit in the same instructions anyway); only
            // where it applies its flux counts.  Its flux Jacobian is the interface's tape entry.
            double u0, Fr, Fy;
            float fp[4];
            arz_trivial_fast((double)ls.x, (double)ls.y, pre, kc, u0, Fr, Fy, fp);
            if (triv) FX[ic] = make_double2(Fr, Fy);
            // append to the lane's queue: the compiler turns the per-lane atomic into ONE LDS atomic per wavefront (add of the
            // number of active lanes) and hands every lane the old value plus its rank among the active ones
            const bool nt = vc & !triv;
            if (nt) Q[atomicAdd(cnt, 1)] = i;
            if (tp && vc) tS[ic] = TapeFp{fp[0], fp[2], fp[3]};
        }
        if (!solve) return;
        lds_only_barrier();
        // ---- phase 2: the queued interfaces ----
        int k0 = tid - (rot << 6);
        if (k0 < 0) k0 += ncell;
        const unsigned i_q = (unsigned)Q[k0 <= N ? k0 : 0];  // read beside the count, not behind it (one LDS round trip less)
        const int qn = *cnt;
        if (k0 < qn) __builtin_amdgcn_s_setprio(3);          // the workgroup's critical path
        if (tp && k0 == 0) { tH[0] = (unsigned)qn; tH[1] = 0u; }
        float4 *tE = tp + geo.s_f4 + geo.h_f4;
        unsigned short *tI = tape_idx(tH, geo);
        for (int k = k0; k < qn; k += ncell) {
            const unsigned i = (k == k0) ? i_q : (unsigned)Q[k];
            const CellRec *lf = CR + i, *rt = CR + i + 1;
            const float4 ls = lf->st, rs = rt->st;
            const double2 lsh = lf->sh, lq0 = lf->q0, rsh = rt->sh;
            CellPre cl, cr;
            cl.s = lsh.x; cl.h = lsh.y; cl.q0 = lq0.x;
            cr.s = rsh.x; cr.h = rsh.y; cr.q0 = 0.;
            Iface f;
            arz_interface_fast_pre((double)ls.x, (double)ls.y, (double)ls.z, (double)ls.w, cl,
                                   (double)rs.x, (double)rs.y, (double)rs.z, (double)rs.w, cr, kc, f);
            FX[i] = make_double2(f.Fr, f.Fy);
            if (tp) {
                tE[2 * k] = make_float4(f.A[0], f.A[1], f.A[2], f.A[3]);
                tE[2 * k + 1] = make_float4(f.B[0], f.B[1], f.B[2], f.B[3]);
                tI[k] = (unsigned short)i;
            }
            if (f.cfl_bad && fault_step < 0) { fault_step = n; fault_index = (int)i; }
        }
        __builtin_amdgcn_s_setprio(0);
        if (tid == 0) CNT[(n + 1) & 1] = 1;
        if (++rot == Wc) rot = 0;
        lds_only_barrier();
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    if (T == 0) {
        body(no{}, no{}, 0);
    } else {
        body(no{}, yes{}, 0);
        for (int n = 1; n < T; ++n) body(yes{}, yes{}, n);
        body(yes{}, no{}, T);
    }
    if (fault_step >= 0) raise_fault(err, DHTS_FAULT_CFL, fault_step, lane, fault_index);
}

// (Round 3's lane-group form of this kernel -- kG lanes per workgroup with one phase-2 list over their queues, 3.39 -> 3.22 ms at
// config 2 -- was what config 2 launched until the pair kernel of round 4 took every shape it covered; no plan selected it any more and
// it was removed in round 6.  docs/history/rounds_1-3_kernel_notebook.md keeps its measurements.)
#ifdef DHTS_FWD3_STAMPS
extern __device__ long long dhts_fwd_clock[2][16][2];     // macro_fwd_pairs.inc (instrumented builds only)
#endif

#include "macro_fwd_pairs.inc"

// ---- reverse sweeps -------------------------------------------------------------------------------------------------
// g' = J^T g per step: grad_cell[a][k] = dqs[a][k]^T g[a]; g'[b] = (c1[b] + c2[b-1]) + c0[b+1]   (dmacro_lane.py:283-303)
// LDS planes of (N + 2) floats: index k + 1 holds cell k; slots 0 and N + 1 stay zero for c2 / c0 so edge cells add 0.
struct BwdPlanes {
    float *Gr, *Gy, *C0r, *C0y, *C2r, *C2y;
};
__device__ __forceinline__ BwdPlanes bwd_planes(float *lds, int P) {
    return BwdPlanes{lds, lds + P, lds + 2 * P, lds + 3 * P, lds + 4 * P, lds + 5 * P};
}
// the cell blocks from the interface products exactly as the forward forms them (dmacro_lane.py:126-129)
__device__ __forceinline__ void cell_blocks(const float4 &aL, const float4 &bL, const float4 &aR, const float4 &bR, float cf, float ncf,
                                            float4 &d0, float4 &d1, float4 &d2) {
    d0.x = ncf * (-aL.x); d0.y = ncf * (-aL.y); d0.z = ncf * (-aL.z); d0.w = ncf * (-aL.w);
    d2.x = ncf * bR.x; d2.y = ncf * bR.y; d2.z = ncf * bR.z; d2.w = ncf * bR.w;
    d1.x = 1.f - cf * (aR.x - bL.x); d1.y = 0.f - cf * (aR.y - bL.y);
    d1.z = 0.f - cf * (aR.z - bL.z); d1.w = 1.f - cf * (aR.w - bL.w);
}

// Reverse sweep over the single-step operator's blocked tape [lane][3][Np][4] (T = 1 from the C ABI; any T works).
// grid = L workgroups of `blockDim.x` threads (multiple of 64); dynamic LDS = 6 * (N + 2) floats.
__global__ void macro_blocks_bwd_kernel(
    int L, int N, int T, const float4 *__restrict__ tape,
    const float *__restrict__ g_r_in, const float *__restrict__ g_y_in, const float *__restrict__ g_hist,
    float *__restrict__ g_r_out, float *__restrict__ g_y_out, double *__restrict__ g_ghost, dhts_error *err) {
    extern __shared__ float lds[];
    const int lane = blockIdx.x;
    const int t = threadIdx.x;
    const int B = blockDim.x;
    const int P = N + 2;
    const BwdPlanes pl = bwd_planes(lds, P);
    float *Gr = pl.Gr, *Gy = pl.Gy, *C0r = pl.C0r, *C0y = pl.C0y, *C2r = pl.C2r, *C2y = pl.C2y;
    const size_t base = (size_t)lane * N;
    const int Np = (N + 63) & ~63;
    const size_t tape_row = (size_t)3 * Np;

    for (int k = t; k < P; k += B) { C0r[k] = 0.f; C0y[k] = 0.f; C2r[k] = 0.f; C2y[k] = 0.f; Gr[k] = 0.f; Gy[k] = 0.f; }
    __syncthreads();
    for (int k = t; k < N; k += B) { Gr[k + 1] = g_r_in[base + k]; Gy[k + 1] = g_y_in[base + k]; }
    __syncthreads();

    double ghl_r = 0., ghl_y = 0., ghr_r = 0., ghr_y = 0.;   // ghost cotangent sums (thread 0 / thread of cell N-1)
    int bad_step = -1, bad_cell = 0;     // first non-finite cotangent this thread meets (the reverse sweep's first = the latest step)
    for (int step = T - 1; step >= 0; --step) {
        const float4 *tp = tape + ((size_t)step * L + lane) * tape_row;
        const float *gh = g_hist ? g_hist + ((size_t)step * L + lane) * 2 * N : nullptr;
        for (int k = t; k < N; k += B) {
            const float4 d0 = tp[k], d1 = tp[Np + k], d2 = tp[2 * Np + k];
            float gr = Gr[k + 1], gy = Gy[k + 1];
            if (gh) { gr += gh[k]; gy += gh[N + k]; }
            const float c0r = dot2(d0.x, gr, d0.z, gy), c0y = dot2(d0.y, gr, d0.w, gy);
            const float c1r = dot2(d1.x, gr, d1.z, gy), c1y = dot2(d1.y, gr, d1.w, gy);
            const float c2r = dot2(d2.x, gr, d2.z, gy), c2y = dot2(d2.y, gr, d2.w, gy);
            // c0 of cell k goes to cell k-1 (slot k), c2 of cell k goes to cell k+1 (slot k+2)
            C0r[k] = c0r; C0y[k] = c0y;
            C2r[k + 2] = c2r; C2y[k + 2] = c2y;
            Gr[k + 1] = c1r; Gy[k + 1] = c1y;
            if (k == 0) { ghl_r += (double)c0r; ghl_y += (double)c0y; }
            if (k == N - 1) { ghr_r += (double)c2r; ghr_y += (double)c2y; }
        }
        __syncthreads();
        for (int k = t; k < N; k += B) {
            // slot k+1 of C2 = c2 of cell k-1 (0 for k = 0); slot k+1 of C0 = c0 of cell k+1 (0 for k = N-1)
            const float c2lr = (k > 0) ? C2r[k + 1] : 0.f, c2ly = (k > 0) ? C2y[k + 1] : 0.f;
            const float c0rr = (k < N - 1) ? C0r[k + 1] : 0.f, c0ry = (k < N - 1) ? C0y[k + 1] : 0.f;
            const float nr = (Gr[k + 1] + c2lr) + c0rr;
            const float ny = (Gy[k + 1] + c2ly) + c0ry;
            Gr[k + 1] = nr; Gy[k + 1] = ny;
            if (bad_step < 0 && !(isfinite(nr) && isfinite(ny))) { bad_step = step; bad_cell = k; }
        }
        __syncthreads();
    }
    for (int k = t; k < N; k += B) { g_r_out[base + k] = Gr[k + 1]; g_y_out[base + k] = Gy[k + 1]; }
    if (g_ghost) {
        if (t == 0) { g_ghost[(size_t)lane * 4 + 0] = ghl_r; g_ghost[(size_t)lane * 4 + 1] = ghl_y; }
        if (t == (N - 1) % B) { g_ghost[(size_t)lane * 4 + 2] = ghr_r; g_ghost[(size_t)lane * 4 + 3] = ghr_y; }
    }
    if (bad_step >= 0) raise_fault(err, DHTS_FAULT_NAN, bad_step, lane, bad_cell);
}

// (A_i, B_i) of interface i from a tape row; SLOT[i] = the exception slot of interface i or 0xffff for a trivial one
// (tape_fill_slots: cleared, then filled from the row's idx list; the caller puts a barrier behind each of the two passes)
__device__ __forceinline__ void tape_iface(const float4 *row, const TapeGeom &geo, int N, int i,
                                           const unsigned short *SLOT, float4 &A, float4 &B) {
    const unsigned s = SLOT[i];
    if (s != 0xffffu) {
        const float4 *E = row + geo.s_f4 + geo.h_f4;
        A = E[2 * s]; B = E[2 * s + 1];
    } else {
        A = tape_trivial_A(reinterpret_cast<const TapeFp *>(row)[i < N ? i : N - 1]);
        B = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void tape_clear_slots(int N, unsigned short *SLOT, int t, int B) {
    for (int i = t; i <= N; i += B) SLOT[i] = 0xffffu;
}
__device__ __forceinline__ void tape_fill_slots(const float4 *row, const TapeGeom &geo, int N, unsigned short *SLOT, int t, int B) {
    const unsigned *H = tape_hdr(row, geo);
    const unsigned short *I = tape_idx(H, geo);
    int cnt = (int)H[0];
    if (cnt > N + 1) cnt = N + 1;
    for (int j = t; j < cnt; j += B) {
        const int i = I[j];
        if (i <= N) SLOT[i] = (unsigned short)j;
    }
}

// Reverse sweep over the rollout tape, one cell per thread (2 <= N <= blockDim.x <= 1024: the rollouts' common shape).  grid = L workgroups of `blockDim.x` threads (multiple of 64).
// Thread k owns cell k, whose blocks are made of the products of interfaces k and k + 1.  Where those are trivial it has them
// in registers (three floats each, read from the row's S block); thread j < cnt also carries exception j to its interface:
// (A, B) into XA / XB [interface], the step's tag into STAMP [interface], one barrier interval before they are used -- a cell
// whose interface carries the step's tag reads the products from there.  The cotangent of the own cell lives in a register;
// what the neighbours contribute goes through C0 / C2.  XA / XB / STAMP / C0 / C2 alternate between two copies with the step
// parity, so a step needs ONE barrier (LDS only: the tape loads stay in flight across it).  The tape entries are loaded
// three steps before they are needed (three register sets trading places, no moves), the exception counts six.  The float32 arithmetic is that of the general kernel below
// (cell_blocks, dot2) two components at a time (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: same operations, same roundings).
// Dynamic LDS: float2 C0[2][N + 2], C2[2][N + 2] (index k + 1 = cell k; slot 1 of C2 and slot N of C0 are never written and
// stay zero: the edge cells add 0) | float4 XA[2][N + 1], XB[2][N + 1] | u32 STAMP[2][N + 2].
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned bits3 __attribute__((ext_vector_type(3)));
typedef unsigned bits4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ TapeFp tape_fp_bits(bits3 b) { TapeFp f; f.f0 = __uint_as_float(b.x); f.f2 = __uint_as_float(b.y); f.f3 = __uint_as_float(b.z); return f; }
__device__ __forceinline__ float4 f4_bits(bits4 b) { return make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)); }
__device__ __forceinline__ v2f pk_dot(v2f dlo, v2f dhi, v2f g) {           // (dot2(dlo.x, g.x, dhi.x, g.y), dot2(dlo.y, g.x, dhi.y, g.y))
    const v2f gx = {g.x, g.x}, gy = {g.y, g.y};
    return __builtin_elementwise_fma(dhi, gy, dlo * gx);
}
// (the planes are sized by the block, kB = 64 .. 512 threads >= N, so that every LDS address is a register plus a literal)
__host__ __device__ inline size_t bwd_fast_lds_bytes(int kB) {
    return 2 * 2 * 8 * (size_t)(kB + 2) + 2 * 2 * 16 * (size_t)(kB + 2) + 2 * 4 * (size_t)(kB + 2);
}
// kHist: per-step cotangents g_hist [T][L][2][N] (a loss on the state history) are added to the cell's cotangent in front of
// every step; they ride in the register sets of the trivial products, three steps ahead.
// kFull: N = kB, every thread holds a cell (BASELINE config 2: 512 cells): no validity masks around the step's pieces.
template <int kB, bool kHist, bool kFull = false>
__global__ __launch_bounds__(kB) __attribute__((amdgpu_waves_per_eu(4, 4))) void macro_rollout_bwd_fast_kernel(     // <= 128 VGPRs
    int L, int N, int T, double cc, const float4 *__restrict__ tape,
    const float *__restrict__ g_r_in, const float *__restrict__ g_y_in, const float *__restrict__ g_hist,
    float *__restrict__ g_r_out, float *__restrict__ g_y_out, double *__restrict__ g_ghost, dhts_error *err) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = blockIdx.x;
    const int t = threadIdx.x;
    constexpr int B = kB;
    constexpr int P = kB + 2;
    float4 *XA = reinterpret_cast<float4 *>(lds), *XB = XA + 2 * P;                   // [copy][P]
    v2f *C0 = reinterpret_cast<v2f *>(XB + 2 * P), *C2 = C0 + 2 * P;                  // [copy][P]
    unsigned *STAMP = reinterpret_cast<unsigned *>(C2 + 2 * P);                       // [copy][P]
    const size_t base = (size_t)lane * N;
    const TapeGeom geo = tape_geom(N);
    const v2f cf = {(float)cc, (float)cc}, ncf = {(float)(-cc), (float)(-cc)};
    const v2f zero2 = {0.f, 0.f}, e0 = {1.f, 0.f}, e1 = {0.f, 1.f};
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int k = t;
    const bool vk = kFull || k < N;
    const unsigned kl = vk ? k : N - 1;
    const unsigned kr = (kl + 1 < (unsigned)N) ? kl + 1 : N - 1;      // interface N is always an exception: its S entry does not exist
    // the wavefronts that hold cell 0 / cell N - 1 sum the ghosts' cotangents (a scalar: the test costs the others one s_cbranch)
    const int edge_wave = __builtin_amdgcn_readfirstlane(((t >> 6) == 0 || (t >> 6) == ((N - 1) >> 6)) ? 1 : 0);

    for (int i = t; i < 2 * P; i += B) { C0[i] = zero2; C2[i] = zero2; STAMP[i] = 0u; }
    v2f g = zero2;
    if (vk) g = v2f{g_r_in[base + k], g_y_in[base + k]};

    double gh_r = 0., gh_y = 0.;         // ghost cotangent sums: thread 0 the left ghost's, thread N - 1 the right one's
    int n_fin = 0;                       // checks of the cell's cotangent that found it finite: a non-finite one stays so (every later value is a
                                         // sum of products with it), so the count says where the first one appeared -- T + 1 checks, newest step first
    // byte offsets inside a row (32-bit, on top of the row's uniform base address)
    const unsigned off_sl = 12u * kl, off_sr = 12u * kr;
    const unsigned off_c = 16u * geo.s_f4;
    const unsigned off_i = 16u * geo.s_f4 + 8u + 2u * t;
    const unsigned off_e = 16u * (geo.s_f4 + geo.h_f4) + 32u * t;
    const size_t row_bytes = 16 * geo.row_f4;
    const char *tb = reinterpret_cast<const char *>(tape) + (size_t)lane * row_bytes;
    const size_t stride = (size_t)L * row_bytes;
    // (plain variables and macros, not structs handed to lambdas: those end up in scratch memory)
    // The two LDS copies are told apart by Q = (T - 1 - step) & 1; the three tape rows a step reads from are running pointers
    // (no multiplications in the loop).
    // per-step cotangents: row of step - 4 as well (the set refilled in the interval of `step` serves step - 4's blocks AND the
    // cotangent added in front of step - 4)
    const size_t hstride = (size_t)L * 2 * N;
    const float *hb = kHist ? g_hist + (size_t)lane * 2 * N + kl : nullptr;
    const float *pH = kHist ? hb + (size_t)(T > 5 ? T - 5 : 0) * hstride : nullptr;
    const char *pS = tb + (size_t)(T > 5 ? T - 5 : 0) * stride;      // row of step - 4 in the interval of `step`
    const char *pE = tb + (size_t)(T > 6 ? T - 6 : 0) * stride;      // row of step - 5
    const char *pC = tb + (size_t)(T > 9 ? T - 9 : 0) * stride;      // row of step - 8
#define DHTS_ROW(step_) (tb + (size_t)((step_) > 0 ? (step_) : 0) * stride)
    // Tape reads are BUFFER loads (round 5): the row's address is a scalar (the running pointer as the descriptor's base), the
    // thread's place in the row a 32-bit register -- no 64-bit vector add per load (the compiler forms one for a global load whose
    // zero-extended offset was hoisted out of the loop: six v_lshl_add_u64 per step), and always the vector memory path (vmcnt).
    const int row_rec = (int)row_bytes;
#define DHTS_RSRC(rb_) __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(rb_), 0, row_rec, 0x00020000)
#define DHTS_LOAD_CNT(rb_, c_) c_ = __builtin_amdgcn_raw_buffer_load_b32(DHTS_RSRC(rb_), off_c, 0, 0);
#define DHTS_LOAD_S(rb_, hp_, sl_, sr_, gh_)                                             \
    {                                                                                    \
        const __amdgpu_buffer_rsrc_t rs_ = DHTS_RSRC(rb_);                               \
        sl_ = tape_fp_bits(__builtin_amdgcn_raw_buffer_load_b96(rs_, off_sl, 0, 0));     \
        sr_ = tape_fp_bits(__builtin_amdgcn_raw_buffer_load_b96(rs_, off_sr, 0, 0));     \
        if (kHist) gh_ = v2f{(hp_)[0], (hp_)[N]};                                        \
    }
#define DHTS_LOAD_E(rb_, c_, ea_, eb_, ix_)                                              \
    if (t < (c_)) {                                                                      \
        const __amdgpu_buffer_rsrc_t rs_ = DHTS_RSRC(rb_);                               \
        ix_ = __builtin_amdgcn_raw_buffer_load_b16(rs_, off_i, 0, 0);                    \
        ea_ = f4_bits(__builtin_amdgcn_raw_buffer_load_b128(rs_, off_e, 0, 0));          \
        eb_ = f4_bits(__builtin_amdgcn_raw_buffer_load_b128(rs_, off_e + 16u, 0, 0));    \
    }
    // exception j of step_ to its interface, in LDS copy Q
#define DHTS_SCATTER(step_, Q, c_, ea_, eb_, ix_)                                        \
    {                                                                                    \
        if ((t < (c_)) & (ix_ <= (unsigned)N)) {                                         \
            XA[(Q) * P + ix_] = ea_; XB[(Q) * P + ix_] = eb_; STAMP[(Q) * P + ix_] = (unsigned)(step_) + 1u; \
        }                                                                                \
        if ((c_) > B) scatter_rest(step_, c_, XA + (Q) * P, XB + (Q) * P, STAMP + (Q) * P); \
    }
    // more exceptions than threads (the one-phase forward kernel flags every interface): the rest, without prefetch
    auto scatter_rest = [&](int step, int cnt, float4 *xa, float4 *xb, unsigned *st) {
        const float4 *row = reinterpret_cast<const float4 *>(DHTS_ROW(step));
        const unsigned short *I = tape_idx(tape_hdr(row, geo), geo);
        const float4 *E = row + geo.s_f4 + geo.h_f4;
        if (cnt > N + 1) cnt = N + 1;
        for (int j = t + B; j < cnt; j += B) {
            const int i = I[j];
            if (i <= N) { xa[i] = E[2 * j]; xb[i] = E[2 * j + 1]; st[i] = (unsigned)step + 1u; }
        }
    };
    // The blocks of step s_ from its products: the trivial ones in (sl_, sr_), the others in LDS copy Q (both stamps are read
    // before either product)
#define DHTS_BLOCKS(s_, Q, sl_, sr_)                                                     \
    {                                                                                    \
        const unsigned tag_ = (unsigned)(s_) + 1u;                                       \
        const unsigned st0_ = STAMP[(Q) * P + k], st1_ = STAMP[(Q) * P + k + 1];         \
        float4 aL = tape_trivial_A(sl_), bL = zero4, aR = tape_trivial_A(sr_), bR = zero4; \
        if (st0_ == tag_) { aL = XA[(Q) * P + k]; bL = XB[(Q) * P + k]; }                \
        if (st1_ == tag_) { aR = XA[(Q) * P + k + 1]; bR = XB[(Q) * P + k + 1]; }        \
        d0lo = ncf * -v2f{aL.x, aL.y}; d0hi = ncf * -v2f{aL.z, aL.w};                    \
        d2lo = ncf * v2f{bR.x, bR.y}; d2hi = ncf * v2f{bR.z, bR.w};                      \
        d1lo = e0 - cf * (v2f{aR.x, aR.y} - v2f{bL.x, bL.y});                            \
        d1hi = e1 - cf * (v2f{aR.z, aR.w} - v2f{bL.z, bL.w});                            \
    }
    // One barrier interval, step s, LDS copy Q (the neighbouring steps use copy R = 1 - Q).  Behind the barrier only what
    // depends on the neighbours: the cell's cotangent after step s + 1 (its own part c1v plus what the two neighbours left in
    // C0 / C2), the three products with the blocks of step s -- formed in the previous interval -- and their hand-over.  Then,
    // off that chain: the blocks of step s - 1 (trivial products in (sl_, sr_), exceptions scattered one interval ago), the
    // exceptions of step s - 2 from (ea_ .. ec_) to LDS, and the refills: exceptions of step s - 5 (their count cq_ arrived three
    // intervals ago), the count of step s - 8, the trivial products of step s - 4.  Everything read from the tape is in flight
    // for three intervals (2 workgroups x 3 steps x 9 KB per CU: what 6 TB/s at ~2 us of latency need).
#define DHTS_STEP(CL, s_, Q, R, sl_, sr_, gh_, ea_, eb_, ix_, ec_, cq_)                  \
    {                                                                                    \
        if (vk) {                                                                        \
            g = (c1v + C2[(R) * P + k + 1]) + C0[(R) * P + k + 1];                       \
            if (kHist) g += gh_cur;                /* the cotangent of the state after step s, if the loss looks at it */ \
            n_fin += (isfinite(g.x) && isfinite(g.y)) ? 1 : 0;                           \
            const v2f c0 = pk_dot(d0lo, d0hi, g), c2v = pk_dot(d2lo, d2hi, g);           \
            c1v = pk_dot(d1lo, d1hi, g);                                                 \
            /* c0 of cell k goes to cell k-1 (slot k), c2 of cell k goes to cell k+1 (slot k+2) */ \
            C0[(Q) * P + k] = c0;                                                        \
            C2[(Q) * P + k + 2] = c2v;                                                   \
            if (edge_wave) {                      /* a scalar branch: two wavefronts of the workgroup take it */ \
                if (k == 0) { gh_r += (double)c0.x; gh_y += (double)c0.y; }              \
                if (k == N - 1) { gh_r += (double)c2v.x; gh_y += (double)c2v.y; }        \
            }                                                                            \
            if (!(CL) || (s_) >= 1) DHTS_BLOCKS((s_) - 1, R, sl_, sr_)                   \
            if (kHist) gh_cur = gh_;               /* of step s - 1: the set is refilled below */ \
        }                                                                                \
        if (!(CL) || (s_) >= 2) DHTS_SCATTER((s_) - 2, Q, ec_, ea_, eb_, ix_);           \
        ec_ = cq_;                                                                       \
        DHTS_LOAD_E(pE, ec_, ea_, eb_, ix_);                                             \
        DHTS_LOAD_CNT(pC, cq_);                                                          \
        DHTS_LOAD_S(pS, pH, sl_, sr_, gh_);                                              \
        if (kHist) pH -= (!(CL) || (s_) > 4) ? hstride : 0;                              \
        pS -= (!(CL) || (s_) > 4) ? stride : 0;                                          \
        pE -= (!(CL) || (s_) > 5) ? stride : 0;                                          \
        pC -= (!(CL) || (s_) > 8) ? stride : 0;                                          \
        lds_only_barrier();                                                              \
    }
    // register set A serves the steps T - 1, T - 4, ..., set B the steps T - 2, T - 5, ..., set C the steps T - 3, T - 6, ...
    TapeFp slA, srA, slB, srB, slC, srC;
    float4 eaA = zero4, ebA = zero4, eaB = zero4, ebB = zero4, eaC = zero4, ebC = zero4;
    unsigned ixA = 0, ixB = 0, ixC = 0;
    int ecA, ecB, ecC, cqA, cqB, cqC;
    v2f d0lo = zero2, d0hi = zero2, d1lo = zero2, d1hi = zero2, d2lo = zero2, d2hi = zero2;
    v2f c1v = g;                                          // "after step T": the incoming cotangent; C0 / C2 are zero
    DHTS_LOAD_CNT(DHTS_ROW(T - 1), ecA);
    DHTS_LOAD_CNT(DHTS_ROW(T - 2), ecB);
    DHTS_LOAD_CNT(DHTS_ROW(T - 3), ecC);
    DHTS_LOAD_CNT(DHTS_ROW(T - 4), cqA);
    DHTS_LOAD_CNT(DHTS_ROW(T - 5), cqB);
    DHTS_LOAD_CNT(DHTS_ROW(T - 6), cqC);
    DHTS_LOAD_E(DHTS_ROW(T - 1), ecA, eaA, ebA, ixA);
    DHTS_LOAD_E(DHTS_ROW(T - 2), ecB, eaB, ebB, ixB);
    DHTS_LOAD_E(DHTS_ROW(T - 3), ecC, eaC, ebC, ixC);
#define DHTS_HROW(step_) (hb + (size_t)((step_) > 0 ? (step_) : 0) * hstride)
    v2f ghA = zero2, ghB = zero2, ghC = zero2, gh_cur = zero2;
    DHTS_LOAD_S(DHTS_ROW(T - 1), DHTS_HROW(T - 1), slA, srA, ghA);
    DHTS_LOAD_S(DHTS_ROW(T - 2), DHTS_HROW(T - 2), slB, srB, ghB);
    DHTS_LOAD_S(DHTS_ROW(T - 3), DHTS_HROW(T - 3), slC, srC, ghC);
    __syncthreads();                                     // the zeroed planes
    DHTS_SCATTER(T - 1, 0, ecA, eaA, ebA, ixA);
    if (T >= 2) DHTS_SCATTER(T - 2, 1, ecB, eaB, ebB, ixB);
    ecA = cqA;
    ecB = cqB;
    DHTS_LOAD_E(DHTS_ROW(T - 4), ecA, eaA, ebA, ixA);
    DHTS_LOAD_E(DHTS_ROW(T - 5), ecB, eaB, ebB, ixB);
    DHTS_LOAD_CNT(DHTS_ROW(T - 7), cqA);
    DHTS_LOAD_CNT(DHTS_ROW(T - 8), cqB);
    lds_only_barrier();
    if (vk) DHTS_BLOCKS(T - 1, 0, slA, srA)
    if (kHist) gh_cur = ghA;                             // of step T - 1
    DHTS_LOAD_S(DHTS_ROW(T - 4), DHTS_HROW(T - 4), slA, srA, ghA);
    lds_only_barrier();                                  // (the first interval scatters step T - 3 into the copy these blocks were read from)
#undef DHTS_HROW
    // on entry to the interval of a step s of class A: the blocks of s are in registers, the exceptions of s - 1 in LDS;
    // (slB, srB) = S(s - 1), set C holds E(s - 2) and cqC the count of s - 5; (slC, srC) = S(s - 2), set A holds E(s - 3) and
    // cqA the count of s - 6; (slA, srA) = S(s - 3), set B holds E(s - 4) and cqB the count of s - 7
    int step = T - 1, q = 0;
    // the main loop: every row a step prefetches from exists (step - 2 > 8), so the running pointers move without end-of-tape
    // clamps and the "is there a step s - 1 / s - 2" tests are literals (round 5: 7 scalar instructions per pointer and step less)
    for (; step >= 11; step -= 3) {
        DHTS_STEP(0, step, q, q ^ 1, slB, srB, ghB, eaC, ebC, ixC, ecC, cqC)
        DHTS_STEP(0, step - 1, q ^ 1, q, slC, srC, ghC, eaA, ebA, ixA, ecA, cqA)
        DHTS_STEP(0, step - 2, q, q ^ 1, slA, srA, ghA, eaB, ebB, ixB, ecB, cqB)
        q ^= 1;
    }
    for (; step >= 2; step -= 3) {
        DHTS_STEP(1, step, q, q ^ 1, slB, srB, ghB, eaC, ebC, ixC, ecC, cqC)
        DHTS_STEP(1, step - 1, q ^ 1, q, slC, srC, ghC, eaA, ebA, ixA, ecA, cqA)
        DHTS_STEP(1, step - 2, q, q ^ 1, slA, srA, ghA, eaB, ebB, ixB, ecB, cqB)
        q ^= 1;
    }
    if (step >= 0) DHTS_STEP(1, step, q, q ^ 1, slB, srB, ghB, eaC, ebC, ixC, ecC, cqC)
    if (step >= 1) DHTS_STEP(1, step - 1, q ^ 1, q, slC, srC, ghC, eaA, ebA, ixA, ecA, cqA)
    if (vk) {                                            // after step 0, whose copy is Q = (T - 1) & 1
        const int oq = ((T - 1) & 1) * P;
        g = (c1v + C2[oq + k + 1]) + C0[oq + k + 1];
        n_fin += (isfinite(g.x) && isfinite(g.y)) ? 1 : 0;
    }
#undef DHTS_BLOCKS
#undef DHTS_ROW
#undef DHTS_LOAD_CNT
#undef DHTS_RSRC
#undef DHTS_LOAD_S
#undef DHTS_LOAD_E
#undef DHTS_SCATTER
#undef DHTS_STEP
    if (vk) { g_r_out[base + k] = g.x; g_y_out[base + k] = g.y; }
    if (g_ghost) {
        if (t == 0) { g_ghost[(size_t)lane * 4 + 0] = gh_r; g_ghost[(size_t)lane * 4 + 1] = gh_y; }
        if (t == N - 1) { g_ghost[(size_t)lane * 4 + 2] = gh_r; g_ghost[(size_t)lane * 4 + 3] = gh_y; }
    }
    if (vk && n_fin < T + 1) raise_fault(err, DHTS_FAULT_NAN, n_fin == 0 ? T - 1 : T - n_fin, lane, k);      // (check i >= 1 looks at the state after step T - i)
}

// The same sweep with TWO cells per thread (cells t and t + kB; kB + 1 < N <= 2 kB): what lanes of 1026 .. 2048 cells take
// (kB = 1024) instead of the general kernel below.  (Measured as a replacement of the one-cell kernel at 1024 lanes x 512 cells
// in round 2 -- four workgroups per CU, every lane resident at once -- it lost, 2.50 against 1.82 ms: the sweep is bound by
// instruction issue and a thread with two cells issues as many instructions per cell as two threads with one.)
// Differences to the kernel above: the exceptions' products stay in the order of their list (XA / XB [slot], slot < kB; a
// row with more exceptions than that leaves the rest in HBM, where the cell that needs one reads it), STAMP [interface] carries
// (step tag << 12 | slot); two register sets instead of three (tape entries two intervals in flight, counts two more), so
// the two LDS copies are literals of the two-step loop body.
// Dynamic LDS: float2 C0[2][P], C2[2][P] | u32 STAMP[2][P] | float4 XA[2][kB], XB[2][kB]      (P = 2 kB + 2)
__host__ __device__ inline size_t bwd_fast2_lds_bytes(int kB) {
    const size_t P = 2 * (size_t)kB + 2;
    return 2 * 2 * 8 * P + 2 * 4 * P + 2 * 2 * 16 * (size_t)kB;
}
template <int kB, bool kFull>          // kFull: N = 2 kB, both cells of every thread exist (one basic block per interval)
__global__ __launch_bounds__(kB) __attribute__((amdgpu_waves_per_eu(4, 4))) void macro_rollout_bwd_fast2_kernel(     // <= 128 VGPRs
    int L, int N, int T, double cc, const float4 *__restrict__ tape,
    const float *__restrict__ g_r_in, const float *__restrict__ g_y_in,
    float *__restrict__ g_r_out, float *__restrict__ g_y_out, double *__restrict__ g_ghost, dhts_error *err) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = blockIdx.x;
    const int t = threadIdx.x;
    constexpr int B = kB;
    constexpr int P = 2 * kB + 2;
    float4 *XA = reinterpret_cast<float4 *>(lds), *XB = XA + 2 * kB;                  // [copy][kB]
    v2f *C0 = reinterpret_cast<v2f *>(XB + 2 * kB), *C2 = C0 + 2 * P;                 // [copy][P]
    unsigned *STAMP = reinterpret_cast<unsigned *>(C2 + 2 * P);                       // [copy][P]
    const size_t base = (size_t)lane * N;
    const TapeGeom geo = tape_geom(N);
    const v2f cf = {(float)cc, (float)cc}, ncf = {(float)(-cc), (float)(-cc)};
    const v2f zero2 = {0.f, 0.f}, e0 = {1.f, 0.f}, e1 = {0.f, 1.f};
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ka = t, kb = t + kB;
    const bool va = kFull || ka < N, vb = kFull || kb < N;
    const unsigned la = va ? ka : N - 1, lb = vb ? kb : N - 1;

    for (int i = t; i < 2 * P; i += B) { C0[i] = zero2; C2[i] = zero2; STAMP[i] = 0u; }
    v2f ga = zero2, gb = zero2;
    if (va) ga = v2f{g_r_in[base + ka], g_y_in[base + ka]};
    if (vb) gb = v2f{g_r_in[base + kb], g_y_in[base + kb]};

    double gh_r = 0., gh_y = 0.;         // ghost cotangent sums: the thread of cell 0 the left ghost's, the thread of cell N - 1 the
                                         // right one's (never the same thread: the launcher keeps N = kB + 1 away from this kernel)
    int bad_step = -1;                   // (first non-finite cotangent: step << 1 | which of the two cells)
    unsigned zv;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zv));     // keeps the count loads on the vector memory path (vmcnt, not lgkmcnt)
    // (the entry right of cell N - 1 does not exist -- interface N is always an exception -- and is never used: its 12 bytes are
    // read from the row's padding / header)
    const unsigned off_la = 12u * la, off_lb = 12u * lb;
    const unsigned off_c = 16u * geo.s_f4 + zv;
    const unsigned off_i = 16u * geo.s_f4 + 8u + 2u * t;
    const unsigned e_base = 16u * (geo.s_f4 + geo.h_f4);
    const unsigned off_e = e_base + 32u * t;
    const size_t row_bytes = 16 * geo.row_f4;
    const char *tb = reinterpret_cast<const char *>(tape) + (size_t)lane * row_bytes;
    const size_t stride = (size_t)L * row_bytes;
    const char *pS = tb + (size_t)(T > 4 ? T - 4 : 0) * stride;      // row of step - 3 in the interval of `step`
    const char *pE = tb + (size_t)(T > 5 ? T - 5 : 0) * stride;      // row of step - 4
    const char *pC = tb + (size_t)(T > 7 ? T - 7 : 0) * stride;      // row of step - 6
#define DHTS_ROW(step_) (tb + (size_t)((step_) > 0 ? (step_) : 0) * stride)
#define DHTS_LOAD_CNT(rb_, c_) c_ = *reinterpret_cast<const int *>((rb_) + off_c);
#define DHTS_LOAD_S(rb_, S_)                                                             \
    {                                                                                    \
        S_##la = *reinterpret_cast<const TapeFp *>((rb_) + off_la);                      \
        S_##ra = *reinterpret_cast<const TapeFp *>((rb_) + off_la + 12);                 \
        S_##lb = *reinterpret_cast<const TapeFp *>((rb_) + off_lb);                      \
        S_##rb = *reinterpret_cast<const TapeFp *>((rb_) + off_lb + 12);                 \
    }
    // (ix_ = interface of exception t, 0xffff where the list is shorter; the list's length rides in its upper half)
#define DHTS_LOAD_E(rb_, c_, ea_, eb_, ix_)                                              \
    ix_ = 0xffffu | ((unsigned)(c_) << 16);                                              \
    if (t < (c_)) {                                                                      \
        ix_ = *reinterpret_cast<const unsigned short *>((rb_) + off_i) | ((unsigned)(c_) << 16); \
        ea_ = *reinterpret_cast<const float4 *>((rb_) + off_e);                          \
        eb_ = *reinterpret_cast<const float4 *>((rb_) + off_e + 16);                     \
    }
    // exception j = t of step_ into copy Q; its interface's stamp points to it.  Exceptions beyond the block size stay in HBM.
#define DHTS_SCATTER(step_, Q, ea_, eb_, ix_)                                            \
    {                                                                                    \
        const unsigned i_ = (ix_) & 0xffffu;                                             \
        if (i_ <= (unsigned)N) {                                                         \
            XA[(Q) * kB + t] = ea_; XB[(Q) * kB + t] = eb_;                              \
            STAMP[(Q) * P + i_] = (((unsigned)(step_) + 1u) << 12) | (unsigned)t;        \
        }                                                                                \
        if ((int)((ix_) >> 16) > B) stamp_rest(step_, (int)((ix_) >> 16), STAMP + (Q) * P); \
    }
    auto stamp_rest = [&](int step, int cnt, unsigned *st) {
        const float4 *row = reinterpret_cast<const float4 *>(DHTS_ROW(step));
        const unsigned short *I = tape_idx(tape_hdr(row, geo), geo);
        if (cnt > N + 1) cnt = N + 1;
        for (int j = t + B; j < cnt; j += B) {
            const int i = I[j];
            if (i <= N) st[i] = (((unsigned)step + 1u) << 12) | (unsigned)j;
        }
    };
    // the products of interface i_ of step s_: exception (slot in LDS copy Q, or beyond the block size: in HBM) or trivial (s3_)
#define DHTS_PRODUCTS(s_, Q, i_, s3_, A_, B_)                                            \
    {                                                                                    \
        const unsigned st_ = STAMP[(Q) * P + (i_)];                                      \
        A_ = tape_trivial_A(s3_); B_ = zero4;                                            \
        if ((st_ >> 12) == (unsigned)(s_) + 1u) {                                        \
            const unsigned j_ = st_ & 4095u;                                             \
            if (j_ < (unsigned)kB) { A_ = XA[(Q) * kB + j_]; B_ = XB[(Q) * kB + j_]; }   \
            else {                                                                       \
                const float4 *E_ = reinterpret_cast<const float4 *>(DHTS_ROW(s_) + e_base); \
                A_ = E_[2 * j_]; B_ = E_[2 * j_ + 1];                                    \
            }                                                                            \
        }                                                                                \
    }
    // the blocks of cell k_ (suffix X_ = a / b) at step s_
#define DHTS_BLOCKS(s_, Q, k_, sl_, sr_, X_)                                             \
    {                                                                                    \
        float4 aL, bL, aR, bR;                                                           \
        DHTS_PRODUCTS(s_, Q, k_, sl_, aL, bL)                                            \
        DHTS_PRODUCTS(s_, Q, (k_) + 1, sr_, aR, bR)                                      \
        d0lo##X_ = ncf * -v2f{aL.x, aL.y}; d0hi##X_ = ncf * -v2f{aL.z, aL.w};            \
        d2lo##X_ = ncf * v2f{bR.x, bR.y}; d2hi##X_ = ncf * v2f{bR.z, bR.w};              \
        d1lo##X_ = e0 - cf * (v2f{aR.x, aR.y} - v2f{bL.x, bL.y});                        \
        d1hi##X_ = e1 - cf * (v2f{aR.z, aR.w} - v2f{bL.z, bL.w});                        \
    }
    // behind the barrier: the cell's cotangent after step s + 1, its three products with the blocks of step s, their hand-over
#define DHTS_CELL(s_, Q, R, k_, X_)                                                      \
    {                                                                                    \
        g##X_ = (c1##X_ + C2[(R) * P + (k_) + 1]) + C0[(R) * P + (k_) + 1];              \
        if (bad_step < 0 && !(isfinite(g##X_.x) && isfinite(g##X_.y))) bad_step = ((((s_) + 1 < T) ? (s_) + 1 : T - 1) << 1) | ((k_) == kb); \
        const v2f c0 = pk_dot(d0lo##X_, d0hi##X_, g##X_), c2v = pk_dot(d2lo##X_, d2hi##X_, g##X_); \
        c1##X_ = pk_dot(d1lo##X_, d1hi##X_, g##X_);                                      \
        C0[(Q) * P + (k_)] = c0;                                                         \
        C2[(Q) * P + (k_) + 2] = c2v;                                                    \
        if (__builtin_amdgcn_ballot_w64(((k_) == 0) | ((k_) == N - 1))) {                \
            asm volatile("" ::: "memory");          /* a real branch: two wavefronts of the workgroup take it */ \
            if ((k_) == 0) { gh_r += (double)c0.x; gh_y += (double)c0.y; }               \
            if ((k_) == N - 1) { gh_r += (double)c2v.x; gh_y += (double)c2v.y; }         \
        }                                                                                \
    }
    // One barrier interval, step s (LDS copy Q, its neighbours' copy R = 1 - Q): the two cells' chain parts, then the blocks of
    // step s - 1 from the trivial products in set S_ and the exceptions scattered one interval ago, the exceptions of step s - 2
    // from (ea_ .. ec_) to LDS, and the refills: exceptions of step s - 4 (count cq_), the count of step s - 6, the trivial
    // products of step s - 3.
#define DHTS_STEP(s_, Q, R, S_, ea_, eb_, ix_, cq_)                                      \
    {                                                                                    \
        if (va) DHTS_CELL(s_, Q, R, ka, a)                                               \
        if (vb) DHTS_CELL(s_, Q, R, kb, b)                                               \
        if ((s_) >= 1) {                                                                 \
            if (va) DHTS_BLOCKS((s_) - 1, R, ka, S_##la, S_##ra, a)                      \
            if (vb) DHTS_BLOCKS((s_) - 1, R, kb, S_##lb, S_##rb, b)                      \
        }                                                                                \
        if ((s_) >= 2) DHTS_SCATTER((s_) - 2, Q, ea_, eb_, ix_);                         \
        DHTS_LOAD_E(pE, cq_, ea_, eb_, ix_);                                             \
        DHTS_LOAD_CNT(pC, cq_);                                                          \
        DHTS_LOAD_S(pS, S_);                                                             \
        pS -= ((s_) > 3) ? stride : 0;                                                   \
        pE -= ((s_) > 4) ? stride : 0;                                                   \
        pC -= ((s_) > 6) ? stride : 0;                                                   \
        lds_only_barrier();                                                              \
    }
    // register set A serves the steps T - 1, T - 3, ..., set B the steps T - 2, T - 4, ...
    TapeFp SAla, SAra, SAlb, SArb, SBla, SBra, SBlb, SBrb;
    float4 eaA = zero4, ebA = zero4, eaB = zero4, ebB = zero4;
    unsigned ixA = 0, ixB = 0;
    int cqA, cqB;
    v2f d0loa = zero2, d0hia = zero2, d1loa = zero2, d1hia = zero2, d2loa = zero2, d2hia = zero2;
    v2f d0lob = zero2, d0hib = zero2, d1lob = zero2, d1hib = zero2, d2lob = zero2, d2hib = zero2;
    v2f c1a = ga, c1b = gb;                               // "after step T": the incoming cotangent; C0 / C2 are zero
    DHTS_LOAD_CNT(DHTS_ROW(T - 1), cqA);
    DHTS_LOAD_CNT(DHTS_ROW(T - 2), cqB);
    DHTS_LOAD_E(DHTS_ROW(T - 1), cqA, eaA, ebA, ixA);
    DHTS_LOAD_E(DHTS_ROW(T - 2), cqB, eaB, ebB, ixB);
    DHTS_LOAD_CNT(DHTS_ROW(T - 3), cqA);
    DHTS_LOAD_CNT(DHTS_ROW(T - 4), cqB);
    DHTS_LOAD_S(DHTS_ROW(T - 1), SA);
    DHTS_LOAD_S(DHTS_ROW(T - 2), SB);
    __syncthreads();                                     // the zeroed planes
    DHTS_SCATTER(T - 1, 0, eaA, ebA, ixA);
    if (T >= 2) DHTS_SCATTER(T - 2, 1, eaB, ebB, ixB);
    DHTS_LOAD_E(DHTS_ROW(T - 3), cqA, eaA, ebA, ixA);
    DHTS_LOAD_E(DHTS_ROW(T - 4), cqB, eaB, ebB, ixB);
    DHTS_LOAD_CNT(DHTS_ROW(T - 5), cqA);
    DHTS_LOAD_CNT(DHTS_ROW(T - 6), cqB);
    lds_only_barrier();
    if (va) DHTS_BLOCKS(T - 1, 0, ka, SAla, SAra, a)
    if (vb) DHTS_BLOCKS(T - 1, 0, kb, SAlb, SArb, b)
    DHTS_LOAD_S(DHTS_ROW(T - 3), SA);
    lds_only_barrier();                                  // (the first interval scatters step T - 3 into the copy these blocks were read from)
    // on entry to the interval of a step s of class A: the blocks of s are in registers, the exceptions of s - 1 in LDS copy 1;
    // set B holds S(s - 1) and E(s - 3), cqB the count of s - 5; set A holds S(s - 2) and E(s - 2), cqA the count of s - 4
    int step = T - 1;
    for (; step >= 1; step -= 2) {
        DHTS_STEP(step, 0, 1, SB, eaA, ebA, ixA, cqA)
        DHTS_STEP(step - 1, 1, 0, SA, eaB, ebB, ixB, cqB)
    }
    if (step == 0) DHTS_STEP(0, 0, 1, SB, eaA, ebA, ixA, cqA)
    {                                                    // after step 0, whose copy is (T - 1) & 1
        const int oq = ((T - 1) & 1) * P;
        if (va) { ga = (c1a + C2[oq + ka + 1]) + C0[oq + ka + 1]; if (bad_step < 0 && !(isfinite(ga.x) && isfinite(ga.y))) bad_step = 0; }
        if (vb) { gb = (c1b + C2[oq + kb + 1]) + C0[oq + kb + 1]; if (bad_step < 0 && !(isfinite(gb.x) && isfinite(gb.y))) bad_step = 1; }
    }
#undef DHTS_ROW
#undef DHTS_LOAD_CNT
#undef DHTS_LOAD_S
#undef DHTS_LOAD_E
#undef DHTS_SCATTER
#undef DHTS_PRODUCTS
#undef DHTS_BLOCKS
#undef DHTS_CELL
#undef DHTS_STEP
    if (va) { g_r_out[base + ka] = ga.x; g_y_out[base + ka] = ga.y; }
    if (vb) { g_r_out[base + kb] = gb.x; g_y_out[base + kb] = gb.y; }
    if (g_ghost) {
        if (ka == 0) { g_ghost[(size_t)lane * 4 + 0] = gh_r; g_ghost[(size_t)lane * 4 + 1] = gh_y; }
        if (ka == N - 1 || kb == N - 1) { g_ghost[(size_t)lane * 4 + 2] = gh_r; g_ghost[(size_t)lane * 4 + 3] = gh_y; }
    }
    if (bad_step >= 0) raise_fault(err, DHTS_FAULT_NAN, bad_step >> 1, lane, (bad_step & 1) ? kb : ka);
}

// Reverse sweep over the rollout tape, any lane length (lanes above 1024 cells, single cells), per-step cotangents (g_hist) or not.
// grid = L workgroups of `blockDim.x` threads (multiple of 64).  Dynamic LDS: 6 planes of (N + 2) floats | u16 SLOT[N + 1].
__global__ __launch_bounds__(512) void macro_rollout_bwd_kernel(
    int L, int N, int T, double cc, const float4 *__restrict__ tape,
    const float *__restrict__ g_r_in, const float *__restrict__ g_y_in, const float *__restrict__ g_hist,
    float *__restrict__ g_r_out, float *__restrict__ g_y_out, double *__restrict__ g_ghost, dhts_error *err) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = blockIdx.x;
    const int t = threadIdx.x;
    const int B = blockDim.x;
    const int P = N + 2;
    const BwdPlanes pl = bwd_planes(lds, P);
    float *Gr = pl.Gr, *Gy = pl.Gy, *C0r = pl.C0r, *C0y = pl.C0y, *C2r = pl.C2r, *C2y = pl.C2y;
    float *xbase = lds + ((6 * P + 3) & ~3);
    const size_t base = (size_t)lane * N;
    const TapeGeom geo = tape_geom(N);
    const float cf = (float)cc, ncf = (float)(-cc);

    for (int k = t; k < P; k += B) { C0r[k] = 0.f; C0y[k] = 0.f; C2r[k] = 0.f; C2y[k] = 0.f; Gr[k] = 0.f; Gy[k] = 0.f; }
    __syncthreads();
    for (int k = t; k < N; k += B) { Gr[k + 1] = g_r_in[base + k]; Gy[k + 1] = g_y_in[base + k]; }
    __syncthreads();

    double ghl_r = 0., ghl_y = 0., ghr_r = 0., ghr_y = 0.;   // ghost cotangent sums (thread 0 / thread of cell N-1)
    int bad_step = -1, bad_cell = 0;     // first non-finite cotangent this thread meets (the reverse sweep's first = the latest step)
    {
        unsigned short *SLOT = reinterpret_cast<unsigned short *>(xbase);
        for (int step = T - 1; step >= 0; --step) {
            const float4 *row = tape + ((size_t)step * L + lane) * geo.row_f4;
            const float *gh = g_hist ? g_hist + ((size_t)step * L + lane) * 2 * N : nullptr;
            tape_clear_slots(N, SLOT, t, B);
            __syncthreads();
            tape_fill_slots(row, geo, N, SLOT, t, B);
            __syncthreads();
            for (int k = t; k < N; k += B) {
                float4 aL, bL, aR, bR, d0, d1, d2;
                tape_iface(row, geo, N, k, SLOT, aL, bL);
                tape_iface(row, geo, N, k + 1, SLOT, aR, bR);
                cell_blocks(aL, bL, aR, bR, cf, ncf, d0, d1, d2);
                float gr = Gr[k + 1], gy = Gy[k + 1];
                if (gh) { gr += gh[k]; gy += gh[N + k]; }
                const float c0r = dot2(d0.x, gr, d0.z, gy), c0y = dot2(d0.y, gr, d0.w, gy);
                const float c1r = dot2(d1.x, gr, d1.z, gy), c1y = dot2(d1.y, gr, d1.w, gy);
                const float c2r = dot2(d2.x, gr, d2.z, gy), c2y = dot2(d2.y, gr, d2.w, gy);
                C0r[k] = c0r; C0y[k] = c0y;
                C2r[k + 2] = c2r; C2y[k + 2] = c2y;
                Gr[k + 1] = c1r; Gy[k + 1] = c1y;
                if (k == 0) { ghl_r += (double)c0r; ghl_y += (double)c0y; }
                if (k == N - 1) { ghr_r += (double)c2r; ghr_y += (double)c2y; }
            }
            __syncthreads();
            for (int k = t; k < N; k += B) {
                const float c2lr = (k > 0) ? C2r[k + 1] : 0.f, c2ly = (k > 0) ? C2y[k + 1] : 0.f;
                const float c0rr = (k < N - 1) ? C0r[k + 1] : 0.f, c0ry = (k < N - 1) ? C0y[k + 1] : 0.f;
                const float nr = (Gr[k + 1] + c2lr) + c0rr;
                const float ny = (Gy[k + 1] + c2ly) + c0ry;
                Gr[k + 1] = nr; Gy[k + 1] = ny;
                if (bad_step < 0 && !(isfinite(nr) && isfinite(ny))) { bad_step = step; bad_cell = k; }
            }
            __syncthreads();
        }
    }
    for (int k = t; k < N; k += B) { g_r_out[base + k] = Gr[k + 1]; g_y_out[base + k] = Gy[k + 1]; }
    if (g_ghost) {
        if (t == 0) { g_ghost[(size_t)lane * 4 + 0] = ghl_r; g_ghost[(size_t)lane * 4 + 1] = ghl_y; }
        if (t == (N - 1) % B) { g_ghost[(size_t)lane * 4 + 2] = ghr_r; g_ghost[(size_t)lane * 4 + 3] = ghr_y; }
    }
    if (bad_step >= 0) raise_fault(err, DHTS_FAULT_NAN, bad_step, lane, bad_cell);
}

// The reference's cell blocks dqs[a][3][2][2] of every (step, lane) from the rollout tape, in the single-step operator's
// layout [step][lane][3][Np][4], formed with the reverse sweep's own code.  grid = T * L workgroups; LDS = u16 SLOT[N + 1].
__global__ void macro_tape_expand_kernel(int N, double cc, const float4 *__restrict__ tape, float4 *__restrict__ dqs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short *SLOT = reinterpret_cast<unsigned short *>(lds);
    const TapeGeom geo = tape_geom(N);
    const int Np = (N + 63) & ~63;
    const float4 *row = tape + (size_t)blockIdx.x * geo.row_f4;
    float4 *out = dqs + (size_t)blockIdx.x * 3 * Np;
    const float cf = (float)cc, ncf = (float)(-cc);
    tape_clear_slots(N, SLOT, threadIdx.x, blockDim.x);
    __syncthreads();
    tape_fill_slots(row, geo, N, SLOT, threadIdx.x, blockDim.x);
    __syncthreads();
    for (int k = threadIdx.x; k < Np; k += blockDim.x) {
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0, d2 = d0;      // the padding is written too (zeros)
        if (k < N) {
            float4 aL, bL, aR, bR;
            tape_iface(row, geo, N, k, SLOT, aL, bL);
            tape_iface(row, geo, N, k + 1, SLOT, aR, bR);
            cell_blocks(aL, bL, aR, bR, cf, ncf, d0, d1, d2);
        }
        out[k] = d0; out[Np + k] = d1; out[2 * Np + k] = d2;
    }
}

// ---- known-answer entry: n independent interfaces, both solver variants ---------------------------------
// in [9][n] double = rL yL uL ueqL rR yR uR ueqR u_max (SoA); variant 0 = production, 1 = reference-order IEEE
__global__ void arz_interface_batch_kernel(int64_t n, int variant, const double *__restrict__ in, double dt, double dx,
                                           int32_t *__restrict__ ci, double *__restrict__ q0, double *__restrict__ flux,
                                           float *__restrict__ dL, float *__restrict__ dR, float *__restrict__ fp,
                                           float *__restrict__ A, float *__restrict__ B, int32_t *__restrict__ cfl_bad,
                                           double *__restrict__ speed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        IfaceConst kc;
        kc.set_um(in[8 * n + i]); kc.set_grid(dt, dx);
        Iface f;
        IfaceDebug dbg;
        if (variant == 1)
            arz_interface_ieee(in[i], in[n + i], in[2 * n + i], in[3 * n + i], in[4 * n + i], in[5 * n + i], in[6 * n + i],
                               in[7 * n + i], kc, f, &dbg);
        else
            arz_interface_fast(in[i], in[n + i], in[2 * n + i], in[3 * n + i], in[4 * n + i], in[5 * n + i], in[6 * n + i],
                               in[7 * n + i], kc, f, &dbg);
        ci[i] = dbg.ci;
        flux[i] = f.Fr; flux[n + i] = f.Fy;
        cfl_bad[i] = f.cfl_bad ? 1 : 0;
        if (speed) { speed[i] = dbg.speed[0]; speed[n + i] = dbg.speed[1]; }
        for (int j = 0; j < 4; ++j) {
            q0[j * n + i] = dbg.q0[j];
            dL[j * n + i] = dbg.dL[j]; dR[j * n + i] = dbg.dR[j]; fp[j * n + i] = dbg.fp[j];
            A[j * n + i] = f.A[j]; B[j * n + i] = f.B[j];
        }
    }
}

// ---- elementwise float32 glue ------------------------------------------------------------------------
__global__ void macro_state_from_ru_kernel(int64_t n, float um, const float *__restrict__ r, const float *__restrict__ u,
                                           float *__restrict__ y, float *__restrict__ q) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float yy, qq;
        glue_from_r_u(r[i], u[i], um, yy, qq);
        y[i] = yy;
        q[i] = qq;
    }
}
__global__ void macro_state_from_ru_bwd_kernel(int64_t n, float um, const float *__restrict__ r, const float *__restrict__ u,
                                               const float *__restrict__ g_y, float *__restrict__ g_r, float *__restrict__ g_u) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gr = g_r[i], gu = 0.f;
        glue_y_bwd(r[i], u[i], um, g_y[i], gr, gu);
        g_r[i] = gr;
        g_u[i] = gu;
    }
}
__global__ void macro_u_tap_bwd_kernel(int64_t n, float um, const float *__restrict__ r, const float *__restrict__ y,
                                       const float *__restrict__ g_u, float *__restrict__ g_r, float *__restrict__ g_y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gr = g_r[i], gy = g_y[i];
        glue_u_bwd(r[i], y[i], um, g_u[i], gr, gy);
        g_r[i] = gr;
        g_y[i] = gy;
    }
}

}  // namespace dhts

// ---- C ABI -----------------------------------------------------------------------------------------------
using namespace dhts;

static inline bool macro_desc_ok(const dhts_macro_desc *d) {
    return d && d->n_lanes > 0 && d->n_cells > 0 && d->n_cells <= DHTS_MACRO_MAX_CELLS && d->dt > 0 && d->dx > 0 && d->u_max > 0;
}
static inline int launch_status() { return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH; }
static inline int grid_1d(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// test / tuning hooks: force the number of wavefronts per lane of the forward kernel (0 = heuristic); select the rollout
// forward kernel (0 = two-phase kernel, 1 = the one-phase kernel the single-step operator uses)
static int dhts_fwd_waves_override = 0;
static int dhts_fwd_variant = 0;
static int dhts_fwd_rotate = 1;        // DHTS_OPT_MACRO_FWD_ROTATE: priority rotation between the two halves of the pair kernel's grid
static int dhts_fwd_group = 0;         // DHTS_OPT_MACRO_FWD_GROUP: traffic lanes per workgroup in the two-phase forward kernel (0 = heuristic)
static inline int padded64(int n) { return (n + 63) & ~63; }

// Wavefronts per lane of the two-phase forward kernel: two 64-cell passes per wavefront whenever the lane fits 16 of them -- the
// two passes of a thread are independent instruction streams, and at 1024 x 512 the four-wave workgroups of all lanes are
// resident at once (4 per CU) instead of taking two rounds.  Measured (tools/sweep_fwd_waves.py, forward ms for 1 / 2 / 4 / 8 /
// 16 waves per lane; literal pass counts 1, 2, run-time above):
//   1024 x 512: 7.5 / 3.95 / 3.50 / 4.07 / -      4096 x 256: 9.3 / 7.50 / 7.93 / - / -      256 x 2048: 19.6 / 10.7 / 6.0 / 4.09 / 3.54
static void macro_fwd2_plan(int N, int &W, int &p) {
    W = dhts_fwd_waves_override > 0 ? dhts_fwd_waves_override : (N + 127) / 128;
    if (W > 16) W = 16;
    if (W < 1) W = 1;
    p = (N + 64 * W - 1) / (64 * W);
    W = (N + 64 * p - 1) / (64 * p);
}
static inline bool macro_fwd2_fits(const dhts_macro_desc *d) {
    return d && sizeof(CellRec) * (size_t)(d->n_cells + 2) + 16 * (size_t)(d->n_cells + 1) + sizeof(int) * (size_t)(d->n_cells + 2) + 16 <=
                    160 * 1024;
}
// The pair kernel (macro_fwd_pairs.inc) takes full lanes of 128 W cells without a state history; traffic lanes per workgroup as
// for the lane-group kernel (four up to three wavefronts per lane, else two; one where the launch would leave CUs without a
// workgroup or the lanes do not divide).  0 = not this kernel (DHTS_OPT_MACRO_FWD_VARIANT = 2 turns it off).
static inline int macro_fwd3_group(const dhts_macro_desc *d, bool want_hist) {
    if (dhts_fwd_variant != 0 || want_hist || d->n_cells < 128 || d->n_cells % 128 != 0 || d->n_cells > 1024) return 0;
    const int W = d->n_cells / 128;
    if (dhts_fwd_waves_override > 0 && dhts_fwd_waves_override != W) return 0;       // (a forced wave count means the lane kernel)
    // lanes per workgroup: four for lanes of up to three wavefronts, one for four (1024 x 512: 2.59 against 2.65 ms for two,
    // 512 x 512: 0.93 / 1.10), two above (profiles/r04k_*, r04q_*)
    int G = dhts_fwd_group > 0 ? dhts_fwd_group : (W <= 3 ? 4 : (W == 4 ? 1 : 2));
    while (G > 1 && (d->n_lanes % G != 0 || (dhts_fwd_group == 0 && d->n_lanes / G < 256) || 64 * W * G > 1024 ||
                     (size_t)G * fwd3_region_bytes(d->n_cells) > 160 * 1024))
        G >>= 1;
    if ((size_t)G * fwd3_region_bytes(d->n_cells) > 160 * 1024) return 0;
    return G;
}
static inline int macro_bwd_fast_block(int N) { return N <= 64 ? 64 : (N <= 128 ? 128 : (N <= 256 ? 256 : (N <= 512 ? 512 : 1024))); }
static inline bool macro_bwd_is_fast(int N, int T) { return N >= 2 && N <= 1024 && T > 0; }      // (T = 0: no tape to prefetch from)
// the two-cells-per-thread sweep (1024 threads): lanes of 1026 .. 2048 cells without per-step cotangents (the step tag has 20 bits)
static inline bool macro_bwd_is_fast2(int N, int T, bool want_hist) { return N > 1025 && N <= 2048 && T > 0 && T < (1 << 20) - 1 && !want_hist; }

static int macro_fwd2_launch(const dhts_macro_desc *d, int T,
                             const float *r, const float *y, const float *u, const float *ueq, const float *ghost,
                             float *r_out, float *y_out, float *u_out, float *ueq_out,
                             float *tape, float *hist, dhts_error *err, void *stream) {
    if (!macro_desc_ok(d) || T < 0 || !r || !y || !u || !ueq || !ghost || !r_out || !y_out || !u_out || !ueq_out)
        return DHTS_E_INVALID;
    const int N = d->n_cells;
    const size_t lds = sizeof(CellRec) * (size_t)(N + 2) + 16 * (size_t)(N + 1) + sizeof(int) * (size_t)(N + 2) + 16;
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    if (lds > 64 * 1024 &&
        (hipFuncSetAttribute((const void *)macro_rollout_fwd2_kernel<1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
         hipFuncSetAttribute((const void *)macro_rollout_fwd2_kernel<2, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
         hipFuncSetAttribute((const void *)macro_rollout_fwd2_kernel<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
         hipFuncSetAttribute((const void *)macro_rollout_fwd2_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
         hipFuncSetAttribute((const void *)macro_rollout_fwd2_kernel<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess))
        return DHTS_E_LAUNCH;
    int W, p;
    macro_fwd2_plan(N, W, p);
#define DHTS_FWD2_ARGS d->n_lanes, N, T, p, d->dt, d->dx, d->u_max, r, y, u, ueq, ghost, r_out, y_out, u_out, ueq_out, \
                       reinterpret_cast<float4 *>(tape), hist, err
    const int G3 = macro_fwd3_group(d, hist != nullptr);
    if (G3 > 0) {
        // the pair kernel: a thread owns two adjacent cells and their right interfaces (macro_fwd_pairs.inc)
        const size_t ldsg = (size_t)G3 * fwd3_region_bytes(N);
#define DHTS_FWD3_ARGS d->n_lanes, N, T, d->dt, d->dx, d->u_max, r, y, u, ueq, ghost, r_out, y_out, u_out, ueq_out, reinterpret_cast<float4 *>(tape), err, dhts_fwd_rotate
#define DHTS_FWDG_ARGS d->n_lanes, N, T, d->dt, d->dx, d->u_max, r, y, u, ueq, ghost, r_out, y_out, u_out, ueq_out, reinterpret_cast<float4 *>(tape), err
#define DHTS_FWD3(GG, TT)                                                                                                         \
    {                                                                                                                             \
        if (ldsg > 64 * 1024 && hipFuncSetAttribute((const void *)macro_rollout_fwd3_kernel<GG, TT>,                              \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg) != hipSuccess)         \
            return DHTS_E_LAUNCH;                                                                                                 \
        macro_rollout_fwd3_kernel<GG, TT><<<d->n_lanes / GG, 64 * GG * (N / 128), ldsg, (hipStream_t)stream>>>(DHTS_FWD3_ARGS);    \
    }
        if (G3 == 1 && tape) DHTS_FWD3(1, true)
        else if (G3 == 1) DHTS_FWD3(1, false)
        else if (G3 == 2 && tape) DHTS_FWD3(2, true)
        else if (G3 == 2) DHTS_FWD3(2, false)
        else if (tape) DHTS_FWD3(4, true)
        else DHTS_FWD3(4, false)
#undef DHTS_FWD3
#undef DHTS_FWDG_ARGS
    } else if (p == 1 && N == 64 * W && hist == nullptr)
        macro_rollout_fwd2_kernel<1, true, false><<<d->n_lanes, 64 * W, lds, (hipStream_t)stream>>>(DHTS_FWD2_ARGS);
    else if (p == 2 && N == 128 * W && hist == nullptr)
        macro_rollout_fwd2_kernel<2, true, false><<<d->n_lanes, 64 * W, lds, (hipStream_t)stream>>>(DHTS_FWD2_ARGS);
    else if (p == 1)
        macro_rollout_fwd2_kernel<1, false, true><<<d->n_lanes, 64 * W, lds, (hipStream_t)stream>>>(DHTS_FWD2_ARGS);
    else if (p == 2)
        macro_rollout_fwd2_kernel<2, false, true><<<d->n_lanes, 64 * W, lds, (hipStream_t)stream>>>(DHTS_FWD2_ARGS);
    else
        macro_rollout_fwd2_kernel<0, false, true><<<d->n_lanes, 64 * W, lds, (hipStream_t)stream>>>(DHTS_FWD2_ARGS);
#undef DHTS_FWD2_ARGS
    return launch_status();
}

template <bool kIface>
static int macro_fwd_launch(const dhts_macro_desc *d, int T,
                            const float *r, const float *y, const float *u, const float *ueq, const float *ghost,
                            float *r_out, float *y_out, float *u_out, float *ueq_out,
                            float *tape, float *hist, dhts_error *err, void *stream) {
    if (!macro_desc_ok(d) || T < 0 || !r || !y || !u || !ueq || !ghost || !r_out || !y_out || !u_out || !ueq_out)
        return DHTS_E_INVALID;
    const int N = d->n_cells;
    const size_t lds = sizeof(float) * 8 * (size_t)(N + 2);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)macro_rollout_fwd_kernel<kIface>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    // Waves per lane: about 5 wavefronts on every SIMD of the chip (256 CUs x 4 SIMDs) when there are few lanes, 2 per lane
    // when there are many, at most 8, and never more than the lane has 63-cell chunks (p = passes per wave, chunk = 64 p - 1
    // cells).  Measured forward times in ms for waves per lane 1 / 2 / 3 / 5 / 8 (tools/sweep_fwd_waves.py, 1000 steps):
    //   4096 x 256: 13.3 / 12.5 / 12.6 / 14.3 / 14.1      2048 x 512: 15.1 / 10.9 / 11.8 / 12.5 / 12.6
    //   1024 x 512:  7.9 /  6.1 /  6.4 /  5.9 /  6.0       512 x 1000: 14.4 / 10.2 /  7.9 /  5.2 /  4.7
    //    256 x 2048: 27.6 / 14.9 /  9.6 /  7.0 /  6.5
    int W = dhts_fwd_waves_override > 0 ? dhts_fwd_waves_override
                                        : (d->n_lanes >= 1536 ? 2 : (int)((5 * 1024 + d->n_lanes - 1) / d->n_lanes));
    if (W > 8) W = 8;
    if (W < 1) W = 1;
    int p = 1;
    while ((N + (64 * p - 1) - 1) / (64 * p - 1) > W) ++p;
    W = (N + (64 * p - 1) - 1) / (64 * p - 1);
    macro_rollout_fwd_kernel<kIface><<<d->n_lanes, 64 * W, lds, (hipStream_t)stream>>>(
        d->n_lanes, N, T, p, d->dt, d->dx, d->u_max, r, y, u, ueq, ghost, r_out, y_out, u_out, ueq_out,
        reinterpret_cast<float4 *>(tape), hist, err);
    return launch_status();
}
static int macro_blocks_bwd_launch(const dhts_macro_desc *d, int T, const float *tape,
                                  const float *g_r, const float *g_y, const float *g_hist,
                                  float *g_r_out, float *g_y_out, double *g_ghost, dhts_error *err, void *stream) {
    if (!macro_desc_ok(d) || T < 0 || (T > 0 && !tape) || !g_r || !g_y || !g_r_out || !g_y_out) return DHTS_E_INVALID;
    const size_t lds = sizeof(float) * 6 * (size_t)(d->n_cells + 2);
    int B = padded64(d->n_cells);
    if (B > 512) B = 512;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)macro_blocks_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    macro_blocks_bwd_kernel<<<d->n_lanes, B, lds, (hipStream_t)stream>>>(
        d->n_lanes, d->n_cells, T, reinterpret_cast<const float4 *>(tape), g_r, g_y, g_hist, g_r_out, g_y_out, g_ghost, err);
    return launch_status();
}
static int macro_rollout_bwd_launch(const dhts_macro_desc *d, int T, const float *tape,
                                    const float *g_r, const float *g_y, const float *g_hist,
                                    float *g_r_out, float *g_y_out, double *g_ghost, dhts_error *err, void *stream) {
    if (!macro_desc_ok(d) || T < 0 || (T > 0 && !tape) || !g_r || !g_y || !g_r_out || !g_y_out) return DHTS_E_INVALID;
    const int N = d->n_cells;
    int B = padded64(N);
    if (B > 512) B = 512;
    if (macro_bwd_is_fast(N, T)) {
        const int kB = macro_bwd_fast_block(N);
        const size_t lds = bwd_fast_lds_bytes(kB);
        if (kB == 1024 &&
            (hipFuncSetAttribute((const void *)macro_rollout_bwd_fast_kernel<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds) != hipSuccess ||
             hipFuncSetAttribute((const void *)macro_rollout_bwd_fast_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds) != hipSuccess))
            return DHTS_E_LAUNCH;
        const float4 *tp = reinterpret_cast<const float4 *>(tape);
        const double cc = d->dt / d->dx;
        hipStream_t st = (hipStream_t)stream;
#define DHTS_BWD_FAST(KB)                                                                                                        \
    if (g_hist)                                                                                                                  \
        macro_rollout_bwd_fast_kernel<KB, true><<<d->n_lanes, KB, lds, st>>>(d->n_lanes, N, T, cc, tp, g_r, g_y, g_hist, g_r_out, \
                                                                             g_y_out, g_ghost, err);                            \
    else if (N == KB && KB >= 128 && KB <= 512)                                                                                  \
        macro_rollout_bwd_fast_kernel<KB, false, true><<<d->n_lanes, KB, lds, st>>>(d->n_lanes, N, T, cc, tp, g_r, g_y, g_hist,   \
                                                                                    g_r_out, g_y_out, g_ghost, err);            \
    else                                                                                                                         \
        macro_rollout_bwd_fast_kernel<KB, false><<<d->n_lanes, KB, lds, st>>>(d->n_lanes, N, T, cc, tp, g_r, g_y, g_hist, g_r_out, \
                                                                              g_y_out, g_ghost, err);
        if (kB == 64) { DHTS_BWD_FAST(64) }
        else if (kB == 128) { DHTS_BWD_FAST(128) }
        else if (kB == 256) { DHTS_BWD_FAST(256) }
        else if (kB == 512) { DHTS_BWD_FAST(512) }
        else { DHTS_BWD_FAST(1024) }
#undef DHTS_BWD_FAST
        return launch_status();
    }
    if (macro_bwd_is_fast2(N, T, g_hist != nullptr)) {
        const size_t lds2 = bwd_fast2_lds_bytes(1024);
        const float4 *tp = reinterpret_cast<const float4 *>(tape);
        if (N == 2048) {
            if (hipFuncSetAttribute((const void *)macro_rollout_bwd_fast2_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
                return DHTS_E_LAUNCH;
            macro_rollout_bwd_fast2_kernel<1024, true><<<d->n_lanes, 1024, lds2, (hipStream_t)stream>>>(d->n_lanes, N, T, d->dt / d->dx, tp, g_r, g_y, g_r_out,
                                                                                                     g_y_out, g_ghost, err);
        } else {
            if (hipFuncSetAttribute((const void *)macro_rollout_bwd_fast2_kernel<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
                return DHTS_E_LAUNCH;
            macro_rollout_bwd_fast2_kernel<1024, false><<<d->n_lanes, 1024, lds2, (hipStream_t)stream>>>(d->n_lanes, N, T, d->dt / d->dx, tp, g_r, g_y, g_r_out,
                                                                                                      g_y_out, g_ghost, err);
        }
        return launch_status();
    }
    const size_t lds = sizeof(float) * (size_t)((6 * (N + 2) + 3) & ~3) + 2 * (size_t)(N + 1);
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)macro_rollout_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    macro_rollout_bwd_kernel<<<d->n_lanes, B, lds, (hipStream_t)stream>>>(
        d->n_lanes, N, T, d->dt / d->dx, reinterpret_cast<const float4 *>(tape), g_r, g_y, g_hist, g_r_out, g_y_out, g_ghost, err);
    return launch_status();
}

extern int dhts_micro_fwd_waves_override;     // micro_kernels.hip

#ifdef DHTS_FWD3_STAMPS
extern "C" int dhts_debug_fwd_clock(long long *out) {         // [2 kernels][16 workgroups][2]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dhts::dhts_fwd_clock), sizeof(long long) * 2 * 16 * 2) == hipSuccess ? 0 : -1;
}
extern "C" int dhts_debug_fwd3_stamps(long long *out) {       // [16 workgroups][16 wavefronts][8], macro_fwd_pairs.inc
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dhts::dhts_fwd3_stamps), sizeof(long long) * 16 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern int dhts_netstep_lds_kb, dhts_netstep_block;      // netstep_hybrid.hip
extern int dhts_hyb_pack;                                // hybrid_kernels.hip
extern int dhts_opt_reward_chain;                        // dhts_common.hip

extern "C" {

int dhts_set_option(int option, int value) {
    if (option == DHTS_OPT_MACRO_FWD_WAVES && value >= 0 && value <= 16) {
        dhts_fwd_waves_override = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_MACRO_FWD_VARIANT && value >= 0 && value <= 2) {
        dhts_fwd_variant = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_MACRO_FWD_ROTATE && (value == 0 || value == 1)) {
        dhts_fwd_rotate = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_NETSTEP_BLOCK && (value == 0 || value == 256 || value == 512 || value == 1024)) {
        dhts_netstep_block = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_NETSTEP_LDS_KB && (value == 0 || (value >= 1 && value <= 158))) {
        dhts_netstep_lds_kb = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_REWARD_CHAIN && (value == 0 || value == 1)) {
        dhts_opt_reward_chain = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_HYB_PACK && value >= 0 && value <= 2) {
        dhts_hyb_pack = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_MACRO_FWD_GROUP && (value == 0 || value == 1 || value == 2 || value == 4)) {
        dhts_fwd_group = value;
        return DHTS_OK;
    }
    if (option == DHTS_OPT_MICRO_FWD_WAVES && (value == 0 || value == 1 || value == 2 || value == 4)) {
        dhts_micro_fwd_waves_override = value;
        return DHTS_OK;
    }
    return DHTS_E_INVALID;
}

int dhts_padded(int n) { return (n + 63) & ~63; }

size_t dhts_macro_tape_bytes(const dhts_macro_desc *d, int T) {
    if (!macro_desc_ok(d) || T < 0) return 0;
    return (size_t)T * d->n_lanes * tape_geom(d->n_cells).row_f4 * sizeof(float4);
}
size_t dhts_macro_step_tape_bytes(const dhts_macro_desc *d) {
    if (!macro_desc_ok(d)) return 0;
    return (size_t)d->n_lanes * 3 * dhts_padded(d->n_cells) * sizeof(float4);
}

int dhts_arz_interface_batch(int64_t n, int variant, const double *in, double dt, double dx, int32_t *case_ind, double *q0,
                             double *flux, float *dL, float *dR, float *fp, float *A, float *B, int32_t *cfl_bad, double *speed,
                             void *stream) {
    if (n < 0 || (variant != 0 && variant != 1) || !in || !case_ind || !q0 || !flux || !dL || !dR || !fp || !A || !B || !cfl_bad)
        return DHTS_E_INVALID;
    if (n == 0) return DHTS_OK;
    arz_interface_batch_kernel<<<grid_1d(n), 256, 0, (hipStream_t)stream>>>(n, variant, in, dt, dx, case_ind, q0, flux, dL, dR, fp,
                                                                          A, B, cfl_bad, speed);
    return launch_status();
}

int dhts_macro_state_from_ru(int64_t n, double u_max, const float *r, const float *u, float *y, float *ueq, void *stream) {
    if (n < 0 || !r || !u || !y || !ueq) return DHTS_E_INVALID;
    if (n == 0) return DHTS_OK;
    macro_state_from_ru_kernel<<<grid_1d(n), 256, 0, (hipStream_t)stream>>>(n, (float)u_max, r, u, y, ueq);
    return launch_status();
}
int dhts_macro_state_from_ru_bwd(int64_t n, double u_max, const float *r, const float *u, const float *g_y,
                                 float *g_r, float *g_u, void *stream) {
    if (n < 0 || !r || !u || !g_y || !g_r || !g_u) return DHTS_E_INVALID;
    if (n == 0) return DHTS_OK;
    macro_state_from_ru_bwd_kernel<<<grid_1d(n), 256, 0, (hipStream_t)stream>>>(n, (float)u_max, r, u, g_y, g_r, g_u);
    return launch_status();
}
int dhts_macro_u_tap_bwd(int64_t n, double u_max, const float *r, const float *y, const float *g_u,
                         float *g_r, float *g_y, void *stream) {
    if (n < 0 || !r || !y || !g_u || !g_r || !g_y) return DHTS_E_INVALID;
    if (n == 0) return DHTS_OK;
    macro_u_tap_bwd_kernel<<<grid_1d(n), 256, 0, (hipStream_t)stream>>>(n, (float)u_max, r, y, g_u, g_r, g_y);
    return launch_status();
}

// rollouts keep the compact tape described above (three floats per trivial interface + the exceptions' products)
int dhts_macro_rollout_fwd(const dhts_macro_desc *d, int T,
                           const float *r, const float *y, const float *u, const float *ueq, const float *ghost,
                           float *r_out, float *y_out, float *u_out, float *ueq_out,
                           float *tape, float *hist, dhts_error *err, void *stream) {
    // lanes whose records do not fit the two-phase kernel's LDS take the one-phase kernel (same tape format)
    if (dhts_fwd_variant == 1 || !macro_fwd2_fits(d))
        return macro_fwd_launch<true>(d, T, r, y, u, ueq, ghost, r_out, y_out, u_out, ueq_out, tape, hist, err, stream);
    return macro_fwd2_launch(d, T, r, y, u, ueq, ghost, r_out, y_out, u_out, ueq_out, tape, hist, err, stream);
}
int dhts_macro_rollout_bwd(const dhts_macro_desc *d, int T, const float *tape,
                           const float *g_r, const float *g_y, const float *g_hist,
                           float *g_r_out, float *g_y_out, double *g_ghost, dhts_error *err, void *stream) {
    return macro_rollout_bwd_launch(d, T, tape, g_r, g_y, g_hist, g_r_out, g_y_out, g_ghost, err, stream);
}
// which kernel instantiations dhts_macro_rollout_fwd / _bwd launch for this shape: the very functions the launches call
int dhts_macro_rollout_plan(const dhts_macro_desc *d, int T, int want_hist, int32_t plan[8]) {
    if (!macro_desc_ok(d) || T < 0 || !plan) return DHTS_E_INVALID;
    for (int k = 0; k < 8; ++k) plan[k] = 0;
    const int N = d->n_cells;
    if (dhts_fwd_variant == 1 || !macro_fwd2_fits(d)) {
        plan[0] = 1;
    } else {
        int W, p;
        macro_fwd2_plan(N, W, p);
        plan[0] = macro_fwd3_group(d, want_hist != 0) > 0 ? 2 : 0;
        plan[1] = W;
        plan[2] = p;
        plan[3] = ((p == 1 || p == 2) && N == 64 * p * W && !want_hist) ? 1 : 0;
    }
    plan[4] = macro_bwd_is_fast(N, T) ? 1 : (macro_bwd_is_fast2(N, T, want_hist != 0) ? 2 : 0);
    plan[5] = plan[4] == 1 ? macro_bwd_fast_block(N) : (plan[4] == 2 ? 1024 : (padded64(N) > 512 ? 512 : padded64(N)));
    plan[6] = want_hist ? 1 : 0;
    plan[7] = plan[0] == 2 ? macro_fwd3_group(d, want_hist != 0) : 1;
    return DHTS_OK;
}
int dhts_macro_tape_expand(const dhts_macro_desc *d, int T, const float *tape, float *dqs, void *stream) {
    if (!macro_desc_ok(d) || T < 0 || (T > 0 && (!tape || !dqs))) return DHTS_E_INVALID;
    if (T == 0) return DHTS_OK;
    const size_t lds = 2 * (size_t)(d->n_cells + 1);
    macro_tape_expand_kernel<<<(unsigned)((size_t)T * d->n_lanes), 256, lds, (hipStream_t)stream>>>(
        d->n_cells, d->dt / d->dx, reinterpret_cast<const float4 *>(tape), reinterpret_cast<float4 *>(dqs));
    return launch_status();
}
// the single-step operator keeps the reference's per-cell blocks dqs[a][3][2][2] (48 B per cell)
int dhts_macro_step_fwd(const dhts_macro_desc *d,
                        const float *r, const float *y, const float *u, const float *ueq, const float *ghost,
                        float *r_out, float *y_out, float *u_out, float *ueq_out,
                        float *tape, dhts_error *err, void *stream) {
    return macro_fwd_launch<false>(d, 1, r, y, u, ueq, ghost, r_out, y_out, u_out, ueq_out, tape, nullptr, err, stream);
}
int dhts_macro_step_bwd(const dhts_macro_desc *d, const float *tape, const float *g_r, const float *g_y,
                        float *g_r_out, float *g_y_out, double *g_ghost, dhts_error *err, void *stream) {
    return macro_blocks_bwd_launch(d, 1, tape, g_r, g_y, nullptr, g_r_out, g_y_out, g_ghost, err, stream);
}

}  // extern "C"
