// network_kernels.hip -- time-fused forward and reverse sweeps of a macro ROAD NETWORK with differentiable traffic
// signals (the itscp environment in `macro` mode), one workgroup per network replica, on gfx950.
//
// What one step is in the reference (example/control/itscp/_env.py:620-768, _simulator.py:56-137,
// road/network/road_network.py:79-111): signals from the action -> every lane's two ghost cells from the time-n state of
// its neighbours, blended between a green and a red value -> one ARZ step per lane -> queue-length loss with a
// running-mean-scaled sigmoid.  Lanes of one network couple every step, so a replica is the atomic unit: it lives in ONE
// workgroup for the whole rollout (state, ghosts and interface results in LDS, one barrier between the phases of a step),
// replicas are independent and fill the chip (256 replicas = one per CU).  HBM sees the per-step state history (needed
// by the loss and ghost adjoints), the Jacobian tape and the loss constants.
//
// Forward phases per step:  signals -> ghosts (2 per lane) -> interface solves (cells + lanes of them, the same device
// code as the straight-lane kernel) -> cell updates + tape -> ordered prefix mean for the loss constants -> lane queues.
// Reverse phases per step:  loss taps -> J^T g per cell -> gather + ghost adjoints (to neighbour edge cells, signals and
// the action), all accumulations in a fixed order (no atomics: results are bitwise repeatable).
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"
#include "arz_device.hpp"

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

struct NetTables {
    const int32_t *lane_ncell, *lane_off, *sig_kind, *inter;   // [L]
    const double *lane_dx;                                      // [L]
    const int32_t *left_src, *left_gate, *right_src;            // [T][L] (per replica when table_stride != 0)
    const double *schedule;                                     // [T][L]
    size_t table_stride;                                        // elements between replicas in the [T][L] tables (0 = shared)
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

// ------------------------------------------------------------------------------------------------------------------
// forward
// LDS (floats): state cur/nxt 2*4*C | ghosts 8*L | iface 12*NI (F as 2 doubles) | sig 2*sq | own 2*L | scan 2*B doubles
// ------------------------------------------------------------------------------------------------------------------
__global__ void net_macro_fwd_kernel(int R, int L, int C, int T, int sq, int F, int n_action, double dt, double um_d,
                                     double static_speed, double veh_len, NetTables tb, const float *__restrict__ action,
                                     float *__restrict__ hist, float4 *__restrict__ tape, float *__restrict__ kc,
                                     float *__restrict__ queue, float *__restrict__ reward, dhts_error *err) {
    extern __shared__ double lds_d[];
    const int rep = blockIdx.x, tid = threadIdx.x, B = blockDim.x;
    const int NI = C + L;
    const int Cp = (C + 63) & ~63;
    // carve LDS: doubles first (alignment)
    double *Fq = lds_d;                                  // [NI][2] flux of Q_0
    double *scan = Fq + 2 * NI;                          // [B] partial sums for the prefix mean
    double *dxd = scan + B;                              // lane cell length [L] (double)
    float *fl = reinterpret_cast<float *>(dxd + L);
    float *S0 = fl;                                      // state buffer 0: [4][C]
    float *S1 = S0 + 4 * C;
    float *G = S1 + 4 * C;                               // ghosts [L][2][4]
    float *AB = G + 8 * L;                               // [NI][8]
    float *sig = AB + 8 * NI;                            // [sq][2]
    float *own = sig + 2 * sq;                           // stored downstream ghost (r, u) of sink lanes [L][2]
    float *ql = own + 2 * L;                             // per-lane queue of this step [L]
    float *dxl_s = ql + L;                               // lane cell length [L]
    int *off_s = reinterpret_cast<int *>(dxl_s + L);     // lane_off [L]
    int *ncl_s = off_s + L;                              // lane_ncell [L]
    int *knd_s = ncl_s + L;                              // sig_kind [L]
    int *int_s = knd_s + L;                              // inter [L]
    int *cell_lane = int_s + L;                          // [C]
    int *iface_lane = cell_lane + C;                     // [NI]
    const float um = (float)um_d;
    const float *act = action + (size_t)rep * n_action;
    const size_t toff = (size_t)rep * tb.table_stride;
    float *hist_r = hist + (size_t)rep * (T + 1) * 4 * C;
    float4 *tape_r = tape + (size_t)rep * T * 3 * Cp;
    float *kc_r = kc + (size_t)rep * T * C;
    float *queue_r = queue + (size_t)rep * T * L;

    IfaceConst kconst;
    kconst.um = um_d; kconst.inv_um = 1.0 / um_d; kconst.inv_15um = 1.0 / (kG1 * um_d); kconst.dt = dt; kconst.dx = 1.0;

    for (int l = tid; l < L; l += B) {
        off_s[l] = tb.lane_off[l]; ncl_s[l] = tb.lane_ncell[l]; knd_s[l] = tb.sig_kind[l]; int_s[l] = tb.inter[l];
        dxl_s[l] = (float)tb.lane_dx[l]; dxd[l] = tb.lane_dx[l];
        for (int i = 0; i < tb.lane_ncell[l]; ++i) cell_lane[tb.lane_off[l] + i] = l;
        for (int k = 0; k <= tb.lane_ncell[l]; ++k) iface_lane[tb.lane_off[l] + l + k] = l;
    }
    // initial state: empty lanes (FullQ(speed_limit): r = y = 0, u = u_eq = u_max)
    for (int c = tid; c < C; c += B) {
        S0[c] = 0.f; S0[C + c] = 0.f; S0[2 * C + c] = um; S0[3 * C + c] = um;
        hist_r[c] = 0.f; hist_r[C + c] = 0.f; hist_r[2 * C + c] = um; hist_r[3 * C + c] = um;
    }
    for (int l = tid; l < L; l += B) { own[2 * l] = 0.f; own[2 * l + 1] = um; }
    double run_sum = 0.;          // running sum / count of the loss samples (uniform across threads)
    long long run_cnt = 0;
    float lane_total = 0.f;       // thread l < L: sum over steps of its lane's queue terms
    int fault_step = -1, fault_lane = 0, fault_index = 0;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        float *cur = (t & 1) ? S1 : S0;
        float *nxt = (t & 1) ? S0 : S1;
        const int32_t *ls_t = tb.left_src + toff + (size_t)t * L;
        const int32_t *lg_t = tb.left_gate + toff + (size_t)t * L;
        const int32_t *rs_t = tb.right_src + toff + (size_t)t * L;
        const double *sc_t = tb.schedule + toff + (size_t)t * L;
        // ---- signals
        for (int k = tid; k < sq; k += B) {
            float we, ns, a, pr; int ai;
            phase_signal(act, n_action, sq, F, t, k, we, ns, a, pr, ai);
            sig[2 * k] = we; sig[2 * k + 1] = ns;
        }
        __syncthreads();
        // ---- ghosts: thread -> (lane, side)   (_simulator.py:56-137)
        for (int j = tid; j < 2 * L; j += B) {
            const int l = j >> 1, side = j & 1;
            float fr, fu, fy, fq;
            if (side == 0) {
                const int ls = ls_t[l], lg = lg_t[l];
                if (ls < 0) {                 // source lane: Python floats in the reference
                    const double gr = sc_t[l];
                    const double gu = um_d * (1. - sqrt(fmax(gr, 0.) + kEps));
                    fr = (float)gr; fu = (float)gu;
                    fy = (float)(gr * (gu - gu)); fq = (float)gu;      // y = r (u - u_eq(r)) = 0 for u = u_eq(r)
                } else {
                    const int last = off_s[ls] + ncl_s[ls] - 1;
                    const float gr = cur[last], gu = cur[2 * C + last];
                    float s;
                    if (lg == -1) s = 0.f;
                    else if (lg == -2) s = 1.f;
                    else { const int kd = knd_s[lg]; s = kd == 0 ? 1.f : sig[2 * int_s[lg] + (kd == 1 ? 0 : 1)]; }
                    fr = gr * s + 0.f * (1.0f - s);
                    fu = gu * s + um * (1.0f - s);
                    glue_from_r_u(fr, fu, um, fy, fq);
                }
            } else {
                const int rs = rs_t[l];
                float gr, gu;
                if (rs < 0) { gr = own[2 * l]; gu = own[2 * l + 1]; }
                else { const int first = off_s[rs]; gr = cur[first]; gu = cur[2 * C + first]; }
                const int kd = knd_s[l];
                const float sg = kd == 0 ? 1.f : sig[2 * int_s[l] + (kd == 1 ? 0 : 1)];
                const float s2 = soft_switch(sg - 0.5f, kSigK);
                fr = s2 * gr + (1.0f - s2) * 1.0f;
                fu = s2 * gu + (1.0f - s2) * 0.0f;
                glue_from_r_u(fr, fu, um, fy, fq);
                own[2 * l] = fr; own[2 * l + 1] = fu;
            }
            float *g = G + (size_t)j * 4;
            g[0] = fr; g[1] = fy; g[2] = fu; g[3] = fq;
        }
        __syncthreads();
        // ---- interface solves: interface id = cell offset + lane index + k, k = 0 .. ncell
        for (int i = tid; i < NI; i += B) {
            const int l = iface_lane[i];                   // lane l owns interfaces [off_l + l, off_l + l + ncell_l]
            const int off = off_s[l], n = ncl_s[l];
            const int k = i - off - l;                     // 0 .. n
            const float *gl = G + (size_t)(2 * l) * 4, *gr_ = G + (size_t)(2 * l + 1) * 4;
            double rL, yL, uL, qL, rR, yR, uR, qR;
            if (k == 0) { rL = gl[0]; yL = gl[1]; uL = gl[2]; qL = gl[3]; }
            else { const int c = off + k - 1; rL = cur[c]; yL = cur[C + c]; uL = cur[2 * C + c]; qL = cur[3 * C + c]; }
            if (k == n) { rR = gr_[0]; yR = gr_[1]; uR = gr_[2]; qR = gr_[3]; }
            else { const int c = off + k; rR = cur[c]; yR = cur[C + c]; uR = cur[2 * C + c]; qR = cur[3 * C + c]; }
            kconst.dx = dxd[l];
            Iface f;
            arz_interface(rL, yL, uL, qL, rR, yR, uR, qR, kconst, f);
            if (f.cfl_bad && fault_step < 0) { fault_step = t; fault_lane = l; fault_index = k; }
            Fq[2 * i] = f.Fr; Fq[2 * i + 1] = f.Fy;
            float *ab = AB + (size_t)i * 8;
            ab[0] = f.A[0]; ab[1] = f.A[1]; ab[2] = f.A[2]; ab[3] = f.A[3];
            ab[4] = f.B[0]; ab[5] = f.B[1]; ab[6] = f.B[2]; ab[7] = f.B[3];
        }
        __syncthreads();
        // ---- cell updates + tape + history   (_macro_lane.py:103-114, dmacro_lane.py:126-129)
        float *hn = hist_r + (size_t)(t + 1) * 4 * C;
        float4 *tp = tape_r + (size_t)t * 3 * Cp;
        for (int c = tid; c < C; c += B) {
            const int l = cell_lane[c];
            const double cc = dt / dxd[l];
            const float cf = (float)cc, ncf = (float)(-cc);
            const int iL = c + l, iR = c + l + 1;
            const float nr = (float)((double)cur[c] + (Fq[2 * iL] - Fq[2 * iR]) * cc);
            const float ny = (float)((double)cur[C + c] + (Fq[2 * iL + 1] - Fq[2 * iR + 1]) * cc);
            float nu, nq;
            glue_from_r_y(nr, ny, um, nu, nq);
            nxt[c] = nr; nxt[C + c] = ny; nxt[2 * C + c] = nu; nxt[3 * C + c] = nq;
            hn[c] = nr; hn[C + c] = ny; hn[2 * C + c] = nu; hn[3 * C + c] = nq;
            const float *aL = AB + (size_t)iL * 8, *aR = AB + (size_t)iR * 8;
            float4 d0, d1, d2;
            d0.x = ncf * (-aL[0]); d0.y = ncf * (-aL[1]); d0.z = ncf * (-aL[2]); d0.w = ncf * (-aL[3]);
            d2.x = ncf * aR[4]; d2.y = ncf * aR[5]; d2.z = ncf * aR[6]; d2.w = ncf * aR[7];
            d1.x = 1.f - cf * (aR[0] - aL[4]); d1.y = 0.f - cf * (aR[1] - aL[5]);
            d1.z = 0.f - cf * (aR[2] - aL[6]); d1.w = 1.f - cf * (aR[3] - aL[7]);
            tp[c] = d0; tp[Cp + c] = d1; tp[2 * Cp + c] = d2;
        }
        __syncthreads();
        // ---- loss constants: k_c = 16 / |mean of all samples (static_speed - u) seen so far, cells in order|
        //      block-wide inclusive prefix sum in double (each thread owns a contiguous chunk of cells)
        const int chunk = (C + B - 1) / B;
        const int c0 = tid * chunk, c1 = min(C, c0 + chunk);
        double part = 0.;
        for (int c = c0; c < c1; ++c) part += (double)((float)static_speed - nxt[2 * C + c]);
        // exclusive scan of the per-thread partial sums: inclusive scan inside each wave by shuffles, then the wave
        // totals (at most 16) are combined through LDS
        double incl = part;
        for (int d = 1; d < 64; d <<= 1) {
            const double up = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d) incl += up;
        }
        const int wv = tid >> 6, nw = B >> 6;
        if ((tid & 63) == 63) scan[wv] = incl;                 // wave totals
        __syncthreads();
        double wave_base = 0., step_total = 0.;
        for (int k = 0; k < nw; ++k) { const double v = scan[k]; if (k < wv) wave_base += v; step_total += v; }
        const double excl = wave_base + (incl - part);
        {
            double acc = run_sum + excl;
            for (int c = c0; c < c1; ++c) {
                acc += (double)((float)static_speed - nxt[2 * C + c]);
                const double mean = acc / (double)(run_cnt + c + 1);
                kc_r[(size_t)t * C + c] = 16.f / fabsf((float)mean);
            }
        }
        run_sum += step_total;
        run_cnt += C;
        __syncthreads();
        // ---- lane queues: q = sum_cells is_static * r dx / len_veh, loss term q^2 dt   (_env.py:664-742)
        for (int l = tid; l < L; l += B) {
            const int off = off_s[l], n = ncl_s[l];
            const float dxl = dxl_s[l];
            float q = 0.f;
            for (int i = 0; i < n; ++i) {
                const int c = off + i;
                const float x = (float)static_speed - nxt[2 * C + c];
                const float is_static = soft_switch(x, kc_r[(size_t)t * C + c]);
                q = q + is_static * (nxt[c] * dxl / (float)veh_len);
            }
            const float term = (q * q) * (float)dt;
            queue_r[(size_t)t * L + l] = term;
            if (l == tid) lane_total = lane_total + (-1.0f) * term;
            else ql[l] = term;          // lanes beyond the block size (L > B): summed by thread 0 below
        }
        __syncthreads();
    }
    // reward = - sum over lanes (outer) and steps (inner) of the queue terms (_env.py:770-797)
    if (tid < L) ql[tid] = lane_total;
    __syncthreads();
    if (tid == 0) {
        float rew = 0.f;
        if (L <= B) {
            for (int l = 0; l < L; ++l) rew = rew + ql[l];
        } else {
            for (int l = 0; l < L; ++l)
                for (int t = 0; t < T; ++t) rew = rew + (-1.0f) * queue_r[(size_t)t * L + l];
        }
        reward[rep] = rew;
    }
    if (fault_step >= 0) net_fault(err, DHTS_FAULT_CFL, fault_step, fault_lane, fault_index);
}

// ------------------------------------------------------------------------------------------------------------------
// reverse
// ------------------------------------------------------------------------------------------------------------------
__global__ void net_macro_bwd_kernel(int R, int L, int C, int T, int sq, int F, int n_action, double dt, double um_d,
                                     double static_speed, double veh_len, NetTables tb, const float *__restrict__ action,
                                     const float *__restrict__ hist, const float4 *__restrict__ tape,
                                     const float *__restrict__ kc, const float *__restrict__ g_reward,
                                     float *__restrict__ g_action, float *__restrict__ own_hist, dhts_error *err) {
    extern __shared__ double lds_d[];
    const int rep = blockIdx.x, tid = threadIdx.x, B = blockDim.x;
    const int Cp = (C + 63) & ~63;
    float *fl = reinterpret_cast<float *>(lds_d);
    float *g = fl;                       // cotangent of (r, y) at time t+1: [2][C]
    float *gp = g + 2 * C;               // ... at time t
    float *c0 = gp + 2 * C;              // dqs[a][0]^T g[a]  -> goes to cell a-1 / the lane's upstream ghost   [2][C]
    float *c2 = c0 + 2 * C;              // dqs[a][2]^T g[a]  -> goes to cell a+1 / the lane's downstream ghost [2][C]
    float *sig = c2 + 2 * C;             // [sq][5]: we, ns, a, prog, (float)a_index
    float *gq = sig + 5 * sq;            // per-lane d reward / d queue [L]
    float *cell_add = gq + L;            // per (lane, side): cotangent for the neighbour's edge cell (r, y) + target  [2L][3]
    float *act_add = cell_add + 6 * L;   // per (lane, side): (value, a_index)  [2L][2]
    float *dxl_s = act_add + 4 * L;      // lane cell length [L]
    int *off_s = reinterpret_cast<int *>(dxl_s + L);
    int *ncl_s = off_s + L;
    int *knd_s = ncl_s + L;
    int *int_s = knd_s + L;
    int *cell_lane = int_s + L;          // [C]
    const float um = (float)um_d;
    const float *act = action + (size_t)rep * n_action;
    const size_t toff = (size_t)rep * tb.table_stride;
    const float *hist_r = hist + (size_t)rep * (T + 1) * 4 * C;
    const float4 *tape_r = tape + (size_t)rep * T * 3 * Cp;
    const float *kc_r = kc + (size_t)rep * T * C;
    float *own_r = own_hist + (size_t)rep * (T + 1) * 2 * L;
    const float gscale = g_reward ? g_reward[rep] : 1.f;

    for (int l = tid; l < L; l += B) {
        off_s[l] = tb.lane_off[l]; ncl_s[l] = tb.lane_ncell[l]; knd_s[l] = tb.sig_kind[l]; int_s[l] = tb.inter[l];
        dxl_s[l] = (float)tb.lane_dx[l];
        for (int i = 0; i < tb.lane_ncell[l]; ++i) cell_lane[tb.lane_off[l] + i] = l;
    }
    // replay the stored downstream ghosts of sink lanes (they depend on themselves and on constants only)
    for (int l = tid; l < L; l += B) { own_r[2 * l] = 0.f; own_r[2 * l + 1] = um; }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int32_t *rs_t = tb.right_src + toff + (size_t)t * L;
        const float *cur = hist_r + (size_t)t * 4 * C;
        for (int l = tid; l < L; l += B) {
            const int rs = rs_t[l];
            const float *o = own_r + (size_t)t * 2 * L;
            float *on = own_r + (size_t)(t + 1) * 2 * L;
            const float gr = rs < 0 ? o[2 * l] : cur[off_s[rs]];
            const float gu = rs < 0 ? o[2 * l + 1] : cur[2 * C + off_s[rs]];
            float we, ns, a, pr; int ai;
            const int kd = knd_s[l];
            float sg = 1.f;
            if (kd != 0) { phase_signal(act, n_action, sq, F, t, int_s[l], we, ns, a, pr, ai); sg = kd == 1 ? we : ns; }
            const float s2 = soft_switch(sg - 0.5f, kSigK);
            on[2 * l] = s2 * gr + (1.0f - s2) * 1.0f;
            on[2 * l + 1] = s2 * gu + (1.0f - s2) * 0.0f;
        }
        __syncthreads();
    }
    for (int c = tid; c < 2 * C; c += B) { g[c] = 0.f; gp[c] = 0.f; }
    double ga = 0.;                      // thread k < n_action: d reward / d action[k]
    bool bad = false;
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const float *cur = hist_r + (size_t)t * 4 * C, *nxt = hist_r + (size_t)(t + 1) * 4 * C;
        const int32_t *ls_t = tb.left_src + toff + (size_t)t * L;
        const int32_t *lg_t = tb.left_gate + toff + (size_t)t * L;
        const int32_t *rs_t = tb.right_src + toff + (size_t)t * L;
        for (int k = tid; k < sq; k += B) {
            float we, ns, a, pr; int ai;
            phase_signal(act, n_action, sq, F, t, k, we, ns, a, pr, ai);
            sig[5 * k] = we; sig[5 * k + 1] = ns; sig[5 * k + 2] = a; sig[5 * k + 3] = pr; sig[5 * k + 4] = (float)ai;
        }
        // ---- (a) loss taps on the state after step t: d reward / d q_l = -2 q_l dt
        for (int l = tid; l < L; l += B) {
            const int off = off_s[l], n = ncl_s[l];
            const float dxl = dxl_s[l];
            float q = 0.f;
            for (int i = 0; i < n; ++i) {
                const int c = off + i;
                const float x = (float)static_speed - nxt[2 * C + c];
                q += soft_switch(x, kc_r[(size_t)t * C + c]) * (nxt[c] * dxl / (float)veh_len);
            }
            gq[l] = gscale * (-1.0f) * (float)dt * 2.f * q;
        }
        __syncthreads();
        for (int c = tid; c < C; c += B) {
            const int l = cell_lane[c];
            const float dxl = dxl_s[l];
            const float rr = nxt[c], yy = nxt[C + c], uu = nxt[2 * C + c];
            const float k = kc_r[(size_t)t * C + c];
            const float x = (float)static_speed - uu;
            const float is_static = soft_switch(x, k);
            const float nveh = rr * dxl / (float)veh_len;
            float gr = g[c] + gq[l] * is_static * (dxl / (float)veh_len);
            float gy = g[C + c];
            glue_u_bwd(rr, yy, um, gq[l] * nveh * (-soft_switch_grad(x, k)), gr, gy);
            // ---- (b) J^T g of this cell   (dmacro_lane.py:283-294)
            const float4 *tp = tape_r + (size_t)t * 3 * Cp;
            const float4 d0 = tp[c], d1 = tp[Cp + c], d2 = tp[2 * Cp + c];
            c0[c] = dot2(d0.x, gr, d0.z, gy); c0[C + c] = dot2(d0.y, gr, d0.w, gy);
            c2[c] = dot2(d2.x, gr, d2.z, gy); c2[C + c] = dot2(d2.y, gr, d2.w, gy);
            gp[c] = dot2(d1.x, gr, d1.z, gy); gp[C + c] = dot2(d1.y, gr, d1.w, gy);
        }
        __syncthreads();
        // ---- (c1) gather inside each lane: g'[b] = (c1[b] + c2[b-1]) + c0[b+1]   (dmacro_lane.py:296-299)
        for (int c = tid; c < C; c += B) {
            const int l = cell_lane[c];
            const int off = off_s[l], n = ncl_s[l];
            float vr = gp[c], vy = gp[C + c];
            if (c > off) { vr += c2[c - 1]; vy += c2[C + c - 1]; }
            if (c < off + n - 1) { vr += c0[c + 1]; vy += c0[C + c + 1]; }
            g[c] = vr; g[C + c] = vy;          // g now holds the lane-internal part of the time-t cotangent
            bad |= !(isfinite(vr) && isfinite(vy));
        }
        // ---- (c2) ghost adjoints: thread -> (lane, side)
        for (int j = tid; j < 2 * L; j += B) {
            const int l = j >> 1, side = j & 1;
            const int off = off_s[l], n = ncl_s[l];
            float tgt = -1.f, add_r = 0.f, add_y = 0.f, a_val = 0.f, a_idx = -1.f;
            if (side == 0) {
                const int ls = ls_t[l], lg = lg_t[l];
                if (ls >= 0) {
                    const int last = off_s[ls] + ncl_s[ls] - 1;
                    const float grn_r = cur[last], grn_u = cur[2 * C + last];
                    float s = 1.f; int kd = 0, it = 0;
                    if (lg == -1) s = 0.f;
                    else if (lg >= 0) { kd = knd_s[lg]; it = int_s[lg]; s = kd == 0 ? 1.f : sig[5 * it + (kd == 1 ? 0 : 1)]; }
                    const float fr = grn_r * s + 0.f * (1.0f - s), fu = grn_u * s + um * (1.0f - s);
                    float g_fr = c0[off], g_fu = 0.f;
                    glue_y_bwd(fr, fu, um, c0[C + off], g_fr, g_fu);
                    add_r = g_fr * s;
                    glue_u_bwd(cur[last], cur[C + last], um, g_fu * s, add_r, add_y);
                    tgt = (float)last;
                    if (lg >= 0 && kd != 0) {
                        const float g_s = g_fr * grn_r + g_fu * (grn_u - um);
                        const float a = sig[5 * it + 2], pr = sig[5 * it + 3];
                        const float dsig = kd == 1 ? soft_switch_grad(a - pr, kSigK) : -soft_switch_grad(pr - a, kSigK);
                        a_val = g_s * dsig; a_idx = sig[5 * it + 4];
                    }
                }
            } else {
                const int rs = rs_t[l];
                const float *o = own_r + (size_t)t * 2 * L;
                const float grn_r = rs < 0 ? o[2 * l] : cur[off_s[rs]];
                const float grn_u = rs < 0 ? o[2 * l + 1] : cur[2 * C + off_s[rs]];
                const int kd = knd_s[l], it = int_s[l];
                const float sg = kd == 0 ? 1.f : sig[5 * it + (kd == 1 ? 0 : 1)];
                const float s2 = soft_switch(sg - 0.5f, kSigK);
                const float fr = s2 * grn_r + (1.0f - s2) * 1.0f, fu = s2 * grn_u + (1.0f - s2) * 0.0f;
                const int lastc = off + n - 1;
                float g_fr = c2[lastc], g_fu = 0.f;
                glue_y_bwd(fr, fu, um, c2[C + lastc], g_fr, g_fu);
                if (rs >= 0) {
                    const int first = off_s[rs];
                    add_r = g_fr * s2;
                    glue_u_bwd(cur[first], cur[C + first], um, g_fu * s2, add_r, add_y);
                    tgt = (float)first;
                }
                if (kd != 0) {
                    const float g_s2 = g_fr * (grn_r - 1.0f) + g_fu * grn_u;
                    const float g_sig = g_s2 * soft_switch_grad(sg - 0.5f, kSigK);
                    const float a = sig[5 * it + 2], pr = sig[5 * it + 3];
                    const float dsig = kd == 1 ? soft_switch_grad(a - pr, kSigK) : -soft_switch_grad(pr - a, kSigK);
                    a_val = g_sig * dsig; a_idx = sig[5 * it + 4];
                }
            }
            cell_add[3 * j] = tgt; cell_add[3 * j + 1] = add_r; cell_add[3 * j + 2] = add_y;
            act_add[2 * j] = a_val; act_add[2 * j + 1] = a_idx;
        }
        __syncthreads();
        // ---- (c3) ordered accumulation (lane id ascending, upstream before downstream): edge cells and the action
        for (int c = tid; c < C; c += B) {
            float vr = g[c], vy = g[C + c];
            // only edge cells of lanes can be targets; scanning 2L entries is cheap and keeps the order fixed
            const int lc = cell_lane[c];
            const bool edge = (c == off_s[lc]) || (c == off_s[lc] + ncl_s[lc] - 1);
            if (edge) {
                for (int j = 0; j < 2 * L; ++j)
                    if (cell_add[3 * j] == (float)c) { vr += cell_add[3 * j + 1]; vy += cell_add[3 * j + 2]; }
            }
            gp[c] = vr; gp[C + c] = vy;
        }
        if (tid < n_action) {
            for (int j = 0; j < 2 * L; ++j)
                if (act_add[2 * j + 1] == (float)tid) ga += (double)act_add[2 * j];
        }
        __syncthreads();
        // time-t cotangent becomes the "next" one of step t-1
        for (int c = tid; c < 2 * C; c += B) g[c] = gp[c];
        __syncthreads();
    }
    if (tid < n_action) g_action[(size_t)rep * n_action + tid] = (float)ga;
    if (bad) net_fault(err, DHTS_FAULT_NAN, 0, 0, tid);
}

}  // namespace dhts

using namespace dhts;

static inline bool net_desc_ok(const dhts_net_desc *d) {
    return d && d->n_replicas > 0 && d->n_lanes > 0 && d->n_cells > 0 && d->n_steps >= 0 && d->n_inter_sq > 0 &&
           d->frames_per_phase > 0 && d->n_action >= d->n_inter_sq && d->n_action <= 1024 && d->dt > 0 && d->u_max > 0 &&
           d->vehicle_length > 0 && (long long)d->n_steps * d->n_cells <= 100000;
}
static inline int net_block(const dhts_net_desc *d) {
    int need = d->n_cells + d->n_lanes;
    if (need < d->n_action) need = d->n_action;
    int B = (need + 63) & ~63;
    return B > 1024 ? 1024 : B;
}
static inline NetTables net_tables(const dhts_net_tables *t) {
    NetTables n;
    n.lane_ncell = t->lane_ncell; n.lane_off = t->lane_off; n.sig_kind = t->sig_kind; n.inter = t->inter; n.lane_dx = t->lane_dx;
    n.left_src = t->left_src; n.left_gate = t->left_gate; n.right_src = t->right_src; n.schedule = t->schedule;
    n.table_stride = (size_t)t->replica_stride;
    return n;
}
static inline bool net_tables_ok(const dhts_net_tables *t) {
    return t && t->lane_ncell && t->lane_off && t->sig_kind && t->inter && t->lane_dx && t->left_src && t->left_gate &&
           t->right_src && t->schedule && t->replica_stride >= 0;
}

extern "C" {

size_t dhts_net_macro_hist_bytes(const dhts_net_desc *d) {
    return net_desc_ok(d) ? sizeof(float) * (size_t)d->n_replicas * (d->n_steps + 1) * 4 * d->n_cells : 0;
}
size_t dhts_net_macro_tape_bytes(const dhts_net_desc *d) {
    return net_desc_ok(d) ? sizeof(float4) * (size_t)d->n_replicas * d->n_steps * 3 * ((d->n_cells + 63) & ~63) : 0;
}

int dhts_net_macro_rollout_fwd(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, float *hist, float *tape,
                               float *kc, float *queue, float *reward, dhts_error *err, void *stream) {
    if (!net_desc_ok(d) || !net_tables_ok(t) || !action || !hist || !tape || !kc || !queue || !reward) return DHTS_E_INVALID;
    const int B = net_block(d), L = d->n_lanes, C = d->n_cells, NI = C + L;
    const size_t lds = sizeof(double) * (2 * (size_t)NI + B + L) +
                       sizeof(float) * (8 * (size_t)C + 8 * L + 8 * (size_t)NI + 2 * d->n_inter_sq + 2 * L + L + L) +
                       sizeof(int) * (4 * (size_t)L + C + NI);
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)net_macro_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    net_macro_fwd_kernel<<<d->n_replicas, B, lds, (hipStream_t)stream>>>(
        d->n_replicas, L, C, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max, d->static_speed,
        d->vehicle_length, net_tables(t), action, hist, reinterpret_cast<float4 *>(tape), kc, queue, reward, err);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

int dhts_net_macro_rollout_bwd(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, const float *hist,
                               const float *tape, const float *kc, const float *g_reward, float *g_action, float *workspace,
                               dhts_error *err, void *stream) {
    if (!net_desc_ok(d) || !net_tables_ok(t) || !action || !hist || !tape || !kc || !g_action || !workspace) return DHTS_E_INVALID;
    const int B = net_block(d), L = d->n_lanes, C = d->n_cells;
    const size_t lds = sizeof(float) * (8 * (size_t)C + 5 * d->n_inter_sq + L + 6 * L + 4 * L + L) + sizeof(int) * (4 * (size_t)L + C) + 64;
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)net_macro_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    net_macro_bwd_kernel<<<d->n_replicas, B, lds, (hipStream_t)stream>>>(
        d->n_replicas, L, C, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max, d->static_speed,
        d->vehicle_length, net_tables(t), action, hist, reinterpret_cast<const float4 *>(tape), kc, g_reward, g_action, workspace, err);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

}  // extern "C"
