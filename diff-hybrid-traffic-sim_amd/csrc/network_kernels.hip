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
#include "net_device.hpp"

namespace dhts {

constexpr long long kNetWindow = 100000;      // RunningMean(100_000), _env.py:122

// bytes of the kernels' LDS carve-ups in front of the staged tables (shared by the kernels and their host wrappers)
__host__ __device__ inline size_t net_fwd_lds_base(int L, int C) {
    const size_t NI = (size_t)C + L;
    return sizeof(double) * (2 * NI + 32) + sizeof(float) * (8 * (size_t)C + 8 * L + 8 * NI + C + L) + sizeof(int) * ((size_t)C + NI);
}
__host__ __device__ inline size_t net_bwd_lds_base(int L, int C, int E, int sq) {
    return sizeof(double) * (((10 * (size_t)C + L + 4 * E + C + 1) / 2 + 1) + 16 * (size_t)sq) + 64;
}
// staged behind the carve-up: the replica's action vector, per lane (first cell | last cell << 16) and (signal kind |
// intersection << 2), the step's signals [sq][4] = (west-east, north-south, d west-east / d a, d north-south / d a)
__host__ __device__ inline size_t net_staged_bytes(int L, int sq, int n_action) {
    return 16 + sizeof(float) * (((size_t)n_action + 3) & ~(size_t)3) + sizeof(int) * 2 * (((size_t)L + 3) & ~(size_t)3) + sizeof(float) * 4 * (size_t)sq;
}
struct NetStaged { float *act; int *lfl; int *linfo; float *sig; };
__device__ __forceinline__ NetStaged net_staged(char *lds, size_t base, int L, int n_action) {
    NetStaged o;
    char *p = lds + ((base + 15) & ~(size_t)15);
    o.act = reinterpret_cast<float *>(p); p += sizeof(float) * ((n_action + 3) & ~3);
    o.lfl = reinterpret_cast<int *>(p); p += sizeof(int) * ((L + 3) & ~3);
    o.linfo = reinterpret_cast<int *>(p); p += sizeof(int) * ((L + 3) & ~3);
    o.sig = reinterpret_cast<float *>(p);
    return o;
}


// ------------------------------------------------------------------------------------------------------------------
// Both kernels run one workgroup per replica with at least C + L threads (C + L <= 1024), so that every role has at most
// one item per thread:  thread i < C + L solves interface i, thread c < C owns cell c, thread j < 2 L owns ghost
// (lane j / 2, side j % 2), thread l < L owns lane l.  Per-step table entries are fetched one step ahead.
// ------------------------------------------------------------------------------------------------------------------

// forward: three barriers per step.  The loss of the state produced by step t-1 is evaluated inside the phases of step t
// (its prefix scan beside the ghost phase, its constants beside the interface solves, its lane sums beside the updates).
// The loss' RunningMean(100 000) (example/common/rms.py:10-19, _env.py:602-606): the mean behind a sample covers the last
// 100 000 samples; beyond that many (T * C > 100 000) a second scan subtracts the samples that have left the window,
// re-read from the state history this kernel writes anyway (sample (step, cell) = static_speed - u of history row step + 1).
// LDS: doubles Fq [NI][2], scanw [2][16] | floats S0, S1 [4][C], G [L][2][4], AB [NI][8], contrib [C], ql [L]
// kHard: an EVALUATION episode (ItscpEnv.step(action, False), what Trainer.evaluate runs, trainer.py:94-142): the signals are
// float(a > progress) (_env.py:928-960), the downstream ghost takes float(signal > 0.5) (_simulator.py:128-137), a cell is static
// when u < static_speed (_env.py:607-617: no running mean, no sigmoid); nothing is kept for a reverse sweep (hist, tape, kc and
// own_hist are not touched and may be NULL).
// kLossWaves: the loss of the previous state is evaluated by wavefronts of its own (threads [Bp, B): one per cell, the first L
// of them also one per lane) beside the physics threads [0, Bp) instead of by the same threads behind their physics role -- the
// two are independent within a phase, and a phase then lasts as long as the longer of them, not as their sum.  Taken when
// pad64(C + L) + pad64(C) threads fit a workgroup.
// kMaxBlock: the launch's block size is at most this (512, 640 or 1024 threads: 256, 168 or 128 vector registers per thread --
// at the 128 of an unbounded kernel the step loop reloaded spilled values from scratch).
template <bool kHard, bool kLossWaves, int kMaxBlock>
__global__ void __launch_bounds__(kMaxBlock) net_macro_fwd_kernel(int R, int L, int C, int T, int sq, int F, int n_action, double dt, double um_d,
                                     double static_speed, double veh_len, NetTables tb, const float *__restrict__ action,
                                     float *__restrict__ hist, float4 *__restrict__ tape, float *__restrict__ kc,
                                     float *__restrict__ queue, float *__restrict__ reward, float *__restrict__ own_hist,
                                     dhts_error *err) {
    extern __shared__ double lds_d[];
    const int rep = blockIdx.x, tid = threadIdx.x, B = blockDim.x;
    const int NI = C + L;
    const int Cp = (C + 63) & ~63;
    double *Fq = lds_d;
    double *scanw = Fq + 2 * NI;
    float *fl = reinterpret_cast<float *>(scanw + 32);
    float *S0 = fl, *S1 = S0 + 4 * C;
    float *G = S1 + 4 * C;
    float *AB = G + 8 * L;
    float *contrib = AB + 8 * NI;
    float *ql = contrib + C;
    int *cell_lane_s = reinterpret_cast<int *>(ql + L);      // [C]  (setup only)
    int *iface_lane_s = cell_lane_s + C;                      // [NI] (setup only)
    const float um = (float)um_d;
    const float s0f = (float)static_speed;
    const NetStaged st_ = net_staged(reinterpret_cast<char *>(lds_d), net_fwd_lds_base(L, C), L, n_action);
    float *act = st_.act; float *sig = st_.sig; int *lfl = st_.lfl, *linfo = st_.linfo;
    for (int i = tid; i < n_action; i += B) act[i] = action[(size_t)rep * n_action + i];
    const size_t toff = (size_t)rep * tb.table_stride;
    float *hist_r = kHard ? nullptr : hist + (size_t)rep * (T + 1) * 4 * C;
    float4 *tape_r = kHard ? nullptr : tape + (size_t)rep * T * 3 * Cp;
    float *kc_r = kHard ? nullptr : kc + (size_t)rep * T * C;
    float *queue_r = queue + (size_t)rep * T * L;
    float *own_w = kHard ? nullptr : own_hist + (size_t)rep * T * 2 * L;       // downstream green values of sink lanes per step (for the reverse)

    if (tid < L) {
        const int off = tb.lane_off[tid], n = tb.lane_ncell[tid];
        for (int i = 0; i < n; ++i) cell_lane_s[off + i] = tid;
        for (int k = 0; k <= n; ++k) iface_lane_s[off + tid + k] = tid;
        lfl[tid] = off | ((off + n - 1) << 16);
        linfo[tid] = tb.sig_kind[tid] | (tb.inter[tid] << 2);
    }
    if (tid < C) {
        S0[tid] = 0.f; S0[C + tid] = 0.f; S0[2 * C + tid] = um; S0[3 * C + tid] = um;          // empty lanes
        if (!kHard) { hist_r[tid] = 0.f; hist_r[C + tid] = 0.f; hist_r[2 * C + tid] = um; hist_r[3 * C + tid] = um; }
    }
    __syncthreads();
    // ---- per-thread roles
    const int Bp = kLossWaves ? B - Cp : 0;         // loss threads start here
    const int lt = kLossWaves ? tid - Bp : tid;     // index of a loss thread (negative: a physics-only thread)
    const bool is_phys = !kLossWaves || tid < Bp;
    const bool is_if = is_phys && tid < NI, is_cell = is_phys && tid < C;
    const bool is_lcell = lt >= 0 && lt < C, is_llane = lt >= 0 && lt < L;
    // ghost threads by side: the left ghosts on threads [0, L), the right ones from the next wavefront boundary on where the
    // physics threads reach that far (a wavefront then runs one side's branch, not both)
    const int gb1 = (((L + 63) & ~63) + L <= (kLossWaves ? Bp : B)) ? ((L + 63) & ~63) : L;
    const bool is_ghost = is_phys && (tid < L || (tid >= gb1 && tid < gb1 + L));
    int i_lane = 0, i_k = 0, i_n = 0, i_off = 0;
    IfaceConst kconst;
    kconst.set_um(um_d); kconst.set_grid(dt, 1.0);
    if (is_if) {
        i_lane = iface_lane_s[tid]; i_off = tb.lane_off[i_lane]; i_n = tb.lane_ncell[i_lane]; i_k = tid - i_off - i_lane;
        kconst.set_grid(dt, tb.lane_dx[i_lane]);
    }
    int c_lane = 0; double c_cc = 0.; float c_dxv = 0.f;
    if (is_cell) { c_lane = cell_lane_s[tid]; c_cc = dt / tb.lane_dx[c_lane]; }
    if (is_lcell) c_dxv = (float)tb.lane_dx[cell_lane_s[lt]] / (float)veh_len;
    const int g_side = tid >= gb1 ? 1 : 0, g_lane = is_ghost ? (g_side ? tid - gb1 : tid) : 0;
    int g_kind = 0, g_inter = 0;
    float own_r = 0.f, own_u = um;                    // stored downstream ghost of a sink lane (side-1 thread)
    if (is_ghost) { g_kind = tb.sig_kind[g_lane]; g_inter = tb.inter[g_lane]; }
    int l_off = 0, l_n = 0;
    if (is_llane) { l_off = tb.lane_off[lt]; l_n = tb.lane_ncell[lt]; }
    __syncthreads();        // setup maps are dead from here on (their LDS is not reused)

    // per-step tables, one step ahead.  Unconditional loads with clamped indices: a load inside a divergent branch merges
    // with the register's old value, and the copy the register allocator places behind it waits for the load at once.
    int p_src = 0, p_gate = 0; double p_sched = 0.;
    const int f_glane = is_ghost ? g_lane : 0;
    const int32_t *f_srcp = g_side == 0 ? tb.left_src : tb.right_src;
    auto fetch = [&](int t) {
        const int tt = t < T ? t : T - 1;
        const size_t o = toff + (size_t)tt * L + f_glane;
        p_src = f_srcp[o]; p_gate = tb.left_gate[o]; p_sched = tb.schedule[o];
    };
    // signals of the step whose ghosts are built next, by the intersection's thread; (phase, frame) are counted
    int sig_ph = 0, sig_fr = 0;
    const int sg_base = (sq <= 64 && B >= 128) ? (((B >> 6) - 1) << 6) : 0;      // signal threads: the last (least loaded) wavefront
    const bool is_sg = tid >= sg_base && tid < sg_base + sq;
    const int sg_q = tid - sg_base;
    auto signals = [&]() {
        if (is_sg) { float we, ns, a, pr; int ai; phase_signal_at(act, n_action, sq, F, sig_ph, sig_fr, sg_q, we, ns, a, pr, ai, kHard); sig[4 * sg_q] = we; sig[4 * sg_q + 1] = ns; }
        if (++sig_fr == F) { sig_fr = 0; ++sig_ph; }
    };
    __syncthreads();        // the staged action vector is complete
    signals();
    __syncthreads();
    if (T > 0) fetch(0);
    double run_in = 0., run_out = 0.; long long run_cnt = 0;      // sums of the samples taken / of those that left the window
    int old_row = -1, old_cell = 0;                                // (history row, cell) of the sample `window` before this thread's next one
    float lane_total = 0.f;
    int fault_step = -1, fault_index = 0;

    // loss pieces of the state held in `st` (evaluated for step index ls >= 0)
    double l_incl = 0., l_incl_out = 0.;
    auto loss_scan = [&](const float *st) {          // phase 1: wave-level inclusive scans over the cells in order
        if (kHard) return;                           // (no running mean in an evaluation episode)
        if (kLossWaves && lt < 0) return;
        double a = 0., b = 0.;
        if (is_lcell) {
            a = (double)(s0f - st[2 * C + lt]);      // x = s0 - u
            const long long idx = run_cnt + lt;
            if (idx >= kNetWindow) {
                if (old_row < 0) { const long long j = idx - kNetWindow; old_row = (int)(j / C) + 1; old_cell = (int)(j % C); }
                else ++old_row;                      // idx grows by C per step: the same cell, one history row later
                b = (double)(s0f - __hip_atomic_load(hist_r + (size_t)old_row * 4 * C + 2 * C + old_cell, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT));      // written long ago by another thread: around L1
            }
        }
        l_incl = wave_scan_add(a);
        l_incl_out = wave_scan_add(b);
        if ((lt & 63) == 63) { scanw[lt >> 6] = l_incl; scanw[16 + (lt >> 6)] = l_incl_out; }
    };
    auto loss_consts = [&](const float *st, int ls) { // phase 2: k_c and the cell's contribution to its lane queue
        if (kHard) {                                 // is_static = 1.0 if speed < static_speed else 0.0 (_env.py:607-617)
            if (is_lcell) contrib[lt] = (st[2 * C + lt] < s0f ? 1.f : 0.f) * (st[lt] * c_dxv);
            return;
        }
        if (kLossWaves && lt < 0) return;
        // the wave totals added in order: the sum in front of this wavefront (a scalar, so is the loop) and the sum of all
        const int wv = __builtin_amdgcn_readfirstlane(lt >> 6), nw = (B - Bp) >> 6;
        double base_a = 0., tot_a = 0., base_b = 0., tot_b = 0.;
        for (int k = 0; k < nw; ++k) {
            if (k == wv) { base_a = tot_a; base_b = tot_b; }
            tot_a += scanw[k]; tot_b += scanw[16 + k];
        }
        if (is_lcell) {
            const long long n = run_cnt + lt + 1;
            const double pin = run_in + base_a + l_incl, pout = run_out + base_b + l_incl_out;
            const bool full = n > kNetWindow;         // one IEEE division, operands selected first
            const double mean = (full ? pin - pout : pin) / (full ? (double)kNetWindow : (double)n);
            const float kk = 16.f / fabsf((float)mean);
            kc_r[(size_t)ls * C + lt] = kk;
            contrib[lt] = soft_switch(s0f - st[2 * C + lt], kk) * (st[lt] * c_dxv);
        }
        run_in += tot_a; run_out += tot_b; run_cnt += C;
    };
    auto loss_lanes = [&](int ls) {                   // phase 3: q = sum of the lane's cells, term q^2 dt
        if (is_llane) {
            // q = the lane's cells added in order (float32, as the reference does); the loads go out eight at a time
            float q = 0.f;
            for (int i0 = 0; i0 < l_n; i0 += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = contrib[l_off + (i0 + j < l_n ? i0 + j : l_n - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (i0 + j < l_n) q = q + v[j];
            }
            const float term = (q * q) * (float)dt;
            queue_r[(size_t)ls * L + lt] = term;
            lane_total = lane_total + (-1.0f) * term;
        }
    };

    // Phase timing (-DDHTS_NET_STAMPS builds only: SRC=network_kernels tools/build_variants.sh stamps:"-DDHTS_NET_STAMPS", then
    // DHTS_LIB=.../variants/libdhts_stamps.so python bench.py --workload itscp_macro --steps 1 --warmup 0 --no-cpu-baseline
    // --no-also): replica 0 prints, per wavefront, the cycles per step in front of each stamp (either role) and at the barriers.
#ifdef DHTS_NET_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define NET_STAMP(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long a_ = __builtin_amdgcn_s_memtime(); st_acc[i] += a_ - st_last; st_last = a_; }
#else
#define NET_STAMP(i)
#endif
    // With wavefronts of their own for the loss the step loop exists in two copies, the physics threads' and the loss threads'
    // (each keeps only its own loop-carried state in registers; the barriers pair up by count).
    if (kLossWaves && is_phys) {
        __builtin_amdgcn_s_setprio(1);        // (the physics threads' phases are the longer ones: 0.752 -> 0.746 ms)
        for (int t = 0; t < T; ++t) {
            constexpr int kRole = 1;
#include "net_fwd_step.inc"
        }
    } else if (kLossWaves) {
        for (int t = 0; t < T; ++t) {
            constexpr int kRole = 2;
#include "net_fwd_step.inc"
        }
    } else {
        for (int t = 0; t < T; ++t) {
            constexpr int kRole = 0;
#include "net_fwd_step.inc"
        }
    }
#ifdef DHTS_NET_STAMPS
    if (rep == 0 && (tid & 63) == 0 && T > 0)
        printf("wave %d: ghosts %lld scan %lld | solve %lld consts %lld | update %lld lanes + signals %lld | barriers %lld  (cycles per step)\n", tid >> 6,
               st_acc[0] / T, st_acc[1] / T, st_acc[2] / T, st_acc[3] / T, st_acc[4] / T, st_acc[5] / T, st_acc[6] / T);
#endif
#undef NET_STAMP
    // loss of the final state
    if (T > 0) {
        const float *fin = (T & 1) ? S1 : S0;
        loss_scan(fin);
        __syncthreads();
        loss_consts(fin, T - 1);
        __syncthreads();
        loss_lanes(T - 1);
    }
    // reward = - sum over lanes (outer) and steps (inner) of the queue terms (_env.py:770-797)
    if (is_llane) ql[lt] = lane_total;
    __syncthreads();
    if (tid == 0) {
        float rew = 0.f;
        for (int l = 0; l < L; ++l) rew = rew + ql[l];
        reward[rep] = rew;
    }
    if (fault_step >= 0) net_fault(err, DHTS_FAULT_CFL, fault_step, i_lane, fault_index);
}

// reverse: two barriers per step, global reads (history rows, loss constants, tape, queue terms, tables) fetched one step
// ahead into registers.
// LDS floats: H0, H1 [3][C] (history rows r, y, u; ping-pong) | c0, c2 [2][C] | gq [L] | inL, inF [E][2] | doubles red [16][sq]
// Limits (checked on the host side of the Python layer): a lane has at most 4 upstream and 4 downstream neighbours.
template <int kMaxBlock>
__global__ void __launch_bounds__(kMaxBlock) net_macro_bwd_kernel(int R, int L, int C, int T, int sq, int F, int n_action, double dt, double um_d,
                                     double static_speed, double veh_len, NetTables tb, const float *__restrict__ action,
                                     const float *__restrict__ hist, const float4 *__restrict__ tape,
                                     const float *__restrict__ kc, const float *__restrict__ queue,
                                     const float *__restrict__ g_reward, float *__restrict__ g_action,
                                     const float *__restrict__ own_hist, dhts_error *err) {
    extern __shared__ double lds_d[];
    const int rep = blockIdx.x, tid = threadIdx.x;
    const int Cp = (C + 63) & ~63;
    const int E = tb.n_edges > 0 ? tb.n_edges : 1;
    const int red_off = (10 * C + L + 4 * E + C + 1) / 2 + 1;     // doubles after the float / int region
    float *fl = reinterpret_cast<float *>(lds_d);
    float *H0 = fl, *H1 = H0 + 3 * C;
    float *c0 = H1 + 3 * C, *c2 = c0 + 2 * C;
    float *gq = c2 + 2 * C;
    float *inL = gq + L;                 // [E][2] cotangent for the LAST cell of lane s from the upstream ghost of its k-th next lane
    float *inF = inL + 2 * E;            // [E][2] cotangent for the FIRST cell of lane s from the downstream ghost of its k-th prev lane
    int *cell_lane_s = reinterpret_cast<int *>(inF + 2 * E);
    double *red = lds_d + red_off;       // [16][sq] per-wave partial sums of the action cotangent
    const float um = (float)um_d, s0f = (float)static_speed;
    const NetStaged st_ = net_staged(reinterpret_cast<char *>(lds_d), net_bwd_lds_base(L, C, E, sq), L, n_action);
    float *act = st_.act; float *sig = st_.sig; int *lfl = st_.lfl, *linfo = st_.linfo;
    for (int i = tid; i < n_action; i += blockDim.x) act[i] = action[(size_t)rep * n_action + i];
    const size_t toff = (size_t)rep * tb.table_stride;
    const float *hist_r = hist + (size_t)rep * (T + 1) * 4 * C;
    const float4 *tape_r = tape + (size_t)rep * T * 3 * Cp;
    const float *kc_r = kc + (size_t)rep * T * C;
    const float *queue_r = queue + (size_t)rep * T * L;
    const float *own_r = own_hist + (size_t)rep * T * 2 * L;
    const float gscale = g_reward ? g_reward[rep] : 1.f;
    // ghost threads by side, as in the forward kernel: left ghosts on threads [0, L), right ones from the next wavefront boundary
    const int gb1 = (((L + 63) & ~63) + L <= (int)blockDim.x) ? ((L + 63) & ~63) : L;
    const bool is_cell = tid < C, is_ghost = tid < L || (tid >= gb1 && tid < gb1 + L), is_lane = tid < L;

    if (is_lane) {
        const int off = tb.lane_off[tid], n = tb.lane_ncell[tid];
        for (int i = 0; i < n; ++i) cell_lane_s[off + i] = tid;
        lfl[tid] = off | ((off + n - 1) << 16);
        linfo[tid] = tb.sig_kind[tid] | (tb.inter[tid] << 2);
    }
    for (int k = tid; k < 2 * E; k += blockDim.x) { inL[k] = 0.f; inF[k] = 0.f; }
    for (int k = tid; k < 16 * sq; k += blockDim.x) red[k] = 0.;      // (wavefronts without ghost threads never write theirs)
    __syncthreads();
    int c_lane = 0, c_first = 0, c_last = 0; float c_dxv = 0.f;
    if (is_cell) {
        c_lane = cell_lane_s[tid]; c_first = tb.lane_off[c_lane]; c_last = c_first + tb.lane_ncell[c_lane] - 1;
        c_dxv = (float)tb.lane_dx[c_lane] / (float)veh_len;
    }
    const int g_side = tid >= gb1 ? 1 : 0, g_lane = is_ghost ? (g_side ? tid - gb1 : tid) : 0;
    int g_kind = 0, g_inter = 0, g_off = 0, g_n = 0;
    if (is_ghost) { g_kind = tb.sig_kind[g_lane]; g_inter = tb.inter[g_lane]; g_off = tb.lane_off[g_lane]; g_n = tb.lane_ncell[g_lane]; }

    // ghost thread: inbox position for each possible source lane (the source of a step is one of the lane's static
    // neighbours): upstream ghost of lane n sourced from s in prev(n) -> entry (index of n in next(s)) of inL, likewise inF
    constexpr int kMaxCand = 4, kMaxEnt = 8;
    int cand_src[kMaxCand], cand_pos[kMaxCand], n_cand = 0;
#pragma unroll
    for (int i = 0; i < kMaxCand; ++i) { cand_src[i] = -1; cand_pos[i] = 0; }
    if (is_ghost) {
        const int32_t *my_ptr = g_side == 0 ? tb.prv_ptr : tb.nxt_ptr, *my_idx = g_side == 0 ? tb.prv_idx : tb.nxt_idx;
        const int32_t *o_ptr = g_side == 0 ? tb.nxt_ptr : tb.prv_ptr, *o_idx = g_side == 0 ? tb.nxt_idx : tb.prv_idx;
        for (int e = my_ptr[g_lane]; e < my_ptr[g_lane + 1] && n_cand < kMaxCand; ++e) {
            const int s_lane = my_idx[e];
            int k = o_ptr[s_lane];
            while (o_idx[k] != g_lane) ++k;
#pragma unroll
            for (int i = 0; i < kMaxCand; ++i) if (i == n_cand) { cand_src[i] = s_lane; cand_pos[i] = k; }
            ++n_cand;
        }
    }
    // edge-cell thread: its inbox entries in ascending slot order (2 n for inL entries, 2 m + 1 for inF entries);
    // entry code = inbox index * 2 + (1 for inF)
    int ent[kMaxEnt], n_ent = 0;
#pragma unroll
    for (int i = 0; i < kMaxEnt; ++i) ent[i] = -1;
    if (is_cell && (tid == c_first || tid == c_last)) {
        int a = tb.nxt_ptr[c_lane], a_end = (tid == c_last) ? tb.nxt_ptr[c_lane + 1] : a;
        int b = tb.prv_ptr[c_lane], b_end = (tid == c_first) ? tb.prv_ptr[c_lane + 1] : b;
        while ((a < a_end || b < b_end) && n_ent < kMaxEnt) {
            const int sa = a < a_end ? 2 * tb.nxt_idx[a] : 0x7fffffff;
            const int sb = b < b_end ? 2 * tb.prv_idx[b] + 1 : 0x7fffffff;
            const int code = sa < sb ? 2 * a : 2 * b + 1;
            if (sa < sb) ++a; else ++b;
#pragma unroll
            for (int i = 0; i < kMaxEnt; ++i) if (i == n_ent) ent[i] = code;
            ++n_ent;
        }
    }

    // ---- prefetch registers (data of step t, fetched during step t+1)
    float p_hr = 0.f, p_hy = 0.f, p_hu = 0.f, p_kc = 0.f, p_q = 0.f, p_own_r = 0.f, p_own_u = 0.f;
    float4 p_d0 = make_float4(0, 0, 0, 0), p_d1 = p_d0, p_d2 = p_d0;
    int p_src = 0, p_gate = 0;
    // unconditional loads with clamped indices (see the forward kernel's fetch)
    const int f_cell = is_cell ? tid : 0, f_lane = is_lane ? tid : 0, f_glane = is_ghost ? g_lane : 0;
    const int32_t *f_srcp = g_side == 0 ? tb.left_src : tb.right_src;
    auto fetch = [&](int t) {
        const int tt = t < 0 ? 0 : t;
        const float *h = hist_r + (size_t)tt * 4 * C;
        p_hr = h[f_cell]; p_hy = h[C + f_cell]; p_hu = h[2 * C + f_cell];
        p_kc = kc_r[(size_t)tt * C + f_cell];
        const float4 *tp = tape_r + (size_t)tt * 3 * Cp;
        p_d0 = tp[f_cell]; p_d1 = tp[Cp + f_cell]; p_d2 = tp[2 * Cp + f_cell];
        p_q = queue_r[(size_t)tt * L + (is_cell ? c_lane : f_lane)];      // cells: the queue term of their own lane
        const size_t o = toff + (size_t)tt * L + f_glane;
        p_src = f_srcp[o]; p_gate = tb.left_gate[o];
        p_own_r = own_r[(size_t)tt * 2 * L + 2 * f_glane]; p_own_u = own_r[(size_t)tt * 2 * L + 2 * f_glane + 1];
    };
    // the final state's history row goes straight to LDS
    if (is_cell) {
        const float *h = hist_r + (size_t)T * 4 * C;
        float *Hn = (T & 1) ? H1 : H0;
        Hn[tid] = h[tid]; Hn[C + tid] = h[C + tid]; Hn[2 * C + tid] = h[2 * C + tid];
    }
    __syncthreads();
    if (T > 0) fetch(T - 1);
    int rev_ph = T > 0 ? (T - 1) / F : 0, rev_fr = T > 0 ? (T - 1) % F : 0;      // (t / F, t % F) of the step being reversed
    const int sg_base = (sq <= 64 && (int)blockDim.x >= 128) ? ((((int)blockDim.x >> 6) - 1) << 6) : 0;   // signal threads: last wavefront
    const bool is_sg = tid >= sg_base && tid < sg_base + sq;
    const int sg_q = tid - sg_base;
    float g_r = 0.f, g_y = 0.f;          // cotangent of this thread's cell at time t+1
    double ga = 0.;                      // thread q < sq: d reward / d action[cur_phase * sq + q], flushed when the phase changes
    int cur_phase = -1;
    int bad_step = -1;          // first non-finite cotangent this thread meets (reverse order: the latest step)
    // actions of phases the rollout never reaches get a zero gradient (thread q owns the entries of intersection q)
    if (tid < sq) for (int k = tid; k < n_action; k += sq) g_action[(size_t)rep * n_action + k] = 0.f;

    if (__any(is_ghost)) __builtin_amdgcn_s_setprio(1);      // the wavefronts with ghost threads carry the longest phase (0.722 -> 0.707 ms)
    for (int t = T - 1; t >= 0; --t) {
        float *Hc = (t & 1) ? H1 : H0;           // row t   (state before step t)
        const float *Hn = (t & 1) ? H0 : H1;     // row t+1 (state after step t)
        // working copies of this step's data, then fetch the next step's
        const float w_kc = p_kc, w_q = p_q, w_own_r = p_own_r, w_own_u = p_own_u;
        const float4 d0 = p_d0, d1 = p_d1, d2 = p_d2;
        const int src = p_src, gate = p_gate;
        if (is_cell) { Hc[tid] = p_hr; Hc[C + tid] = p_hy; Hc[2 * C + tid] = p_hu; }
        if (is_sg) {                         // this step's signals and their derivatives w.r.t. the action entry
            float we, ns, a, pr; int ai;
            phase_signal_at(act, n_action, sq, F, rev_ph, rev_fr, sg_q, we, ns, a, pr, ai);
            sig[4 * sg_q] = we; sig[4 * sg_q + 1] = ns;
            // d sigmoid(k x) / d x = s (1 - s) k with the sigmoid values just computed (0 outside the clamp, like the operator)
            const float zs = (a - pr) * kSigK;
            const bool sat = zs < -16.f || zs > 16.f;
            sig[4 * sg_q + 2] = sat ? 0.f : we * (1.f - we) * kSigK; sig[4 * sg_q + 3] = sat ? 0.f : -(ns * (1.f - ns) * kSigK);
        }
        if (t > 0) fetch(t - 1);
        // ---- phase B (no barrier in front: row t + 1 is in LDS since the previous iteration and every cell evaluates its own
        //      lane's loss weight d reward / d q_l = - 2 q_l dt, q_l = sqrt(term / dt)): loss taps on the state after step t,
        //      then J^T g of this cell (dmacro_lane.py:283-294)
        float v_r = 0.f, v_y = 0.f;
        if (is_cell) {
            const int c = tid;
            const float rr = Hn[c], yy = Hn[C + c], uu = Hn[2 * C + c];
            const float x = s0f - uu;
            float is_static, d_static;
            soft_switch_both(x, w_kc, is_static, d_static);
            const float nveh = rr * c_dxv;
            const float gql = gscale * (-1.0f) * (float)dt * 2.f * sqrtf(w_q / (float)dt);
            float gr = g_r + gql * is_static * c_dxv;
            float gy = g_y;
            glue_u_bwd(rr, yy, um, gql * nveh * (-d_static), gr, gy);
            c0[c] = dot2(d0.x, gr, d0.z, gy); c0[C + c] = dot2(d0.y, gr, d0.w, gy);
            c2[c] = dot2(d2.x, gr, d2.z, gy); c2[C + c] = dot2(d2.y, gr, d2.w, gy);
            v_r = dot2(d1.x, gr, d1.z, gy); v_y = dot2(d1.y, gr, d1.w, gy);
        }
        lds_barrier();
        // ---- phase C: gather inside the lane g'[b] = (c1[b] + c2[b-1]) + c0[b+1] (dmacro_lane.py:296-299); ghost adjoints
        if (is_cell) {
            const int c = tid;
            if (c > c_first) { v_r += c2[c - 1]; v_y += c2[C + c - 1]; }
            if (c < c_last) { v_r += c0[c + 1]; v_y += c0[C + c + 1]; }
        }
        float my_aval = 0.f; int my_akey = -1;
        if (is_ghost) {
            float tgt = -1.f, add_r = 0.f, add_y = 0.f, a_val = 0.f; int a_key = -1;
            if (g_side == 0) {
                if (src >= 0) {
                    const int last = lfl[src] >> 16;
                    const float grn_r = Hc[last], grn_u = Hc[2 * C + last];
                    float s = 1.f; int kd = 0, it = 0;
                    if (gate == -1) s = 0.f;
                    else if (gate >= 0) {
                        kd = linfo[gate] & 3; it = linfo[gate] >> 2;
                        if (kd != 0) s = sig[4 * it + (kd - 1)];
                    }
                    const float fr = grn_r * s + 0.f * (1.0f - s), fu = grn_u * s + um * (1.0f - s);
                    float g_fr = c0[g_off], g_fu = 0.f;
                    glue_y_bwd(fr, fu, um, c0[C + g_off], g_fr, g_fu);
                    add_r = g_fr * s;
                    glue_u_bwd(Hc[last], Hc[C + last], um, g_fu * s, add_r, add_y);
                    tgt = (float)last;
                    if (kd != 0) {
                        const float g_s = g_fr * grn_r + g_fu * (grn_u - um);
                        a_val = g_s * sig[4 * it + 2 + (kd - 1)]; a_key = it;
                    }
                }
            } else {
                const int first = src < 0 ? 0 : (lfl[src] & 0xffff);
                const float grn_r = src < 0 ? w_own_r : Hc[first];
                const float grn_u = src < 0 ? w_own_u : Hc[2 * C + first];
                const float sg = g_kind != 0 ? sig[4 * g_inter + (g_kind - 1)] : 1.f;
                const float s2 = soft_switch(sg - 0.5f, kSigK);
                const float fr = s2 * grn_r + (1.0f - s2) * 1.0f, fu = s2 * grn_u + (1.0f - s2) * 0.0f;
                const int lastc = g_off + g_n - 1;
                float g_fr = c2[lastc], g_fu = 0.f;
                glue_y_bwd(fr, fu, um, c2[C + lastc], g_fr, g_fu);
                if (src >= 0) {
                    add_r = g_fr * s2;
                    glue_u_bwd(Hc[first], Hc[C + first], um, g_fu * s2, add_r, add_y);
                    tgt = (float)first;
                }
                if (g_kind != 0) {
                    const float g_s2 = g_fr * (grn_r - 1.0f) + g_fu * grn_u;
                    const float g_sig = g_s2 * soft_switch_grad(sg - 0.5f, kSigK);
                    a_val = g_sig * sig[4 * g_inter + 2 + (g_kind - 1)]; a_key = g_inter;
                }
            }
            // route the cell cotangent to the inbox entry of (source lane, this lane)
            if (tgt >= 0.f) {
                float *box = g_side == 0 ? inL : inF;
#pragma unroll
                for (int i = 0; i < kMaxCand; ++i)
                    if (cand_src[i] == src) { box[2 * cand_pos[i]] = add_r; box[2 * cand_pos[i] + 1] = add_y; }
            }
            my_aval = a_val; my_akey = a_key;
        }
        // action cotangent: per intersection, fixed-shape reduction in double (the wavefront's sum by DPP moves, then the wave
        // partials); only the wavefronts that hold ghost threads have anything to add
        if (__any(is_ghost))
            for (int q = 0; q < sq; ++q) {
                const double v = wave_scan_add((my_akey == q) ? (double)my_aval : 0.0);
                if ((tid & 63) == 63) red[(tid >> 6) * sq + q] = v;
            }
        lds_barrier();
        // ---- phase D: edge cells take their inbox entries in ascending slot order (lane id ascending, upstream ghost of
        //      lane n = slot 2n before downstream ghost of lane m = slot 2m+1), then clear them; action partials are summed
        if (is_cell) {
#pragma unroll
            for (int i = 0; i < kMaxEnt; ++i) {
                if (ent[i] >= 0) {
                    float *box = (ent[i] & 1) ? inF : inL;
                    const int k = ent[i] >> 1;
                    v_r += box[2 * k]; v_y += box[2 * k + 1];
                    box[2 * k] = 0.f; box[2 * k + 1] = 0.f;
                }
            }
            g_r = v_r; g_y = v_y;
            if (bad_step < 0 && !(isfinite(v_r) && isfinite(v_y))) bad_step = t;
        }
        if (tid < sq) {
            double v = 0.;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += red[w * sq + tid];
            // all contributions of step t belong to the phase of step t
            const int lastp = n_action / sq - 1; const int phase = rev_ph > lastp ? lastp : rev_ph;
            if (phase != cur_phase) { if (cur_phase >= 0) g_action[(size_t)rep * n_action + cur_phase * sq + tid] = (float)ga; ga = 0.; cur_phase = phase; }
            ga += v;
        }
        if (rev_fr == 0) { rev_fr = F - 1; --rev_ph; } else --rev_fr;
        // no barrier here: the next step's first phase only writes rows / gq that nobody reads before its own barrier;
        // inbox entries and the reduction scratch are rewritten two barriers later
    }
    if (tid < sq && cur_phase >= 0) g_action[(size_t)rep * n_action + cur_phase * sq + tid] = (float)ga;
    if (bad_step >= 0) net_fault(err, DHTS_FAULT_NAN, bad_step, rep, tid);
}

}  // namespace dhts

using namespace dhts;

static inline bool net_desc_ok(const dhts_net_desc *d) {
    return d && d->n_replicas > 0 && d->n_lanes > 0 && d->n_cells > 0 && d->n_steps >= 0 && d->n_inter_sq > 0 &&
           d->frames_per_phase > 0 && d->n_action >= d->n_inter_sq && d->n_action <= 1024 && d->dt > 0 && d->u_max > 0 &&
           d->vehicle_length > 0 && (long long)d->n_steps * d->n_cells < (1ll << 31) && d->n_cells + d->n_lanes <= 1024 &&
           d->n_lanes <= d->n_cells;
}
static inline int net_block(const dhts_net_desc *d) {
    int need = d->n_cells + d->n_lanes;
    if (need < d->n_action) need = d->n_action;
    return (need + 63) & ~63;
}
// forward kernels: wavefronts of their own for the loss where they fit the workgroup
static inline int net_fwd_block(const dhts_net_desc *d, bool &loss_waves) {
    const int Bp = net_block(d), B = Bp + ((d->n_cells + 63) & ~63);
    loss_waves = B <= 1024;
    return loss_waves ? B : Bp;
}
static inline NetTables net_tables(const dhts_net_tables *t) {
    NetTables n;
    n.lane_ncell = t->lane_ncell; n.lane_off = t->lane_off; n.sig_kind = t->sig_kind; n.inter = t->inter; n.lane_dx = t->lane_dx;
    n.left_src = t->left_src; n.left_gate = t->left_gate; n.right_src = t->right_src; n.schedule = t->schedule;
    n.table_stride = (size_t)t->replica_stride;
    n.nxt_ptr = t->nxt_ptr; n.nxt_idx = t->nxt_idx; n.prv_ptr = t->prv_ptr; n.prv_idx = t->prv_idx; n.n_edges = t->n_edges;
    return n;
}
static inline bool net_tables_ok(const dhts_net_tables *t) {
    return t && t->lane_ncell && t->lane_off && t->sig_kind && t->inter && t->lane_dx && t->left_src && t->left_gate &&
           t->right_src && t->schedule && t->replica_stride >= 0 && t->nxt_ptr && t->nxt_idx && t->prv_ptr && t->prv_idx &&
           t->n_edges >= 0;
}

// dhts_common.hip: the reward as the reference's one float32 chain, lanes outermost (DHTS_OPT_REWARD_CHAIN)
extern int dhts_opt_reward_chain;
int dhts_launch_reward_chain(int R, int T, int L, const float *queue, const int32_t *lane_macro, int hard, double dt, int loss_steps,
                             float *reward, int stride, void *stream);

extern "C" {

size_t dhts_net_macro_hist_bytes(const dhts_net_desc *d) {
    return net_desc_ok(d) ? sizeof(float) * (size_t)d->n_replicas * (d->n_steps + 1) * 4 * d->n_cells : 0;
}
size_t dhts_net_macro_tape_bytes(const dhts_net_desc *d) {
    return net_desc_ok(d) ? sizeof(float4) * (size_t)d->n_replicas * d->n_steps * 3 * ((d->n_cells + 63) & ~63) : 0;
}

int dhts_net_macro_rollout_fwd(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, float *hist, float *tape,
                               float *kc, float *queue, float *reward, float *workspace, dhts_error *err, void *stream) {
    if (!net_desc_ok(d) || !net_tables_ok(t) || !action || !hist || !tape || !kc || !queue || !reward || !workspace)
        return DHTS_E_INVALID;
    bool lw;
    const int B = net_fwd_block(d, lw), L = d->n_lanes, C = d->n_cells;
    const size_t lds = net_fwd_lds_base(L, C) + net_staged_bytes(L, d->n_inter_sq, d->n_action);
    if (lds > 160 * 1024) return DHTS_E_INVALID;
#define DHTS_NET_FWD_ARGS d->n_replicas, L, C, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max, d->static_speed, \
        d->vehicle_length, net_tables(t), action, hist, reinterpret_cast<float4 *>(tape), kc, queue, reward, workspace, err
#define DHTS_NET_FWD_LAUNCH(LW, MB)                                                                                                   \
    {                                                                                                                                 \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)net_macro_fwd_kernel<false, LW, MB>,                                 \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)              \
            return DHTS_E_LAUNCH;                                                                                                     \
        net_macro_fwd_kernel<false, LW, MB><<<d->n_replicas, B, lds, (hipStream_t)stream>>>(DHTS_NET_FWD_ARGS);                       \
    }
    if (lw) {
        if (B <= 512) DHTS_NET_FWD_LAUNCH(true, 512) else if (B <= 640) DHTS_NET_FWD_LAUNCH(true, 640) else DHTS_NET_FWD_LAUNCH(true, 1024)
    } else {
        if (B <= 512) DHTS_NET_FWD_LAUNCH(false, 512) else if (B <= 640) DHTS_NET_FWD_LAUNCH(false, 640) else DHTS_NET_FWD_LAUNCH(false, 1024)
    }
#undef DHTS_NET_FWD_LAUNCH
#undef DHTS_NET_FWD_ARGS
    if (hipGetLastError() != hipSuccess) return DHTS_E_LAUNCH;
    if (dhts_opt_reward_chain) return dhts_launch_reward_chain(d->n_replicas, d->n_steps, L, queue, nullptr, 0, d->dt, 0, reward, 1, stream);
    return DHTS_OK;
}

int dhts_net_macro_rollout_eval(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, float *queue, float *reward,
                                dhts_error *err, void *stream) {
    if (!net_desc_ok(d) || !net_tables_ok(t) || !action || !queue || !reward) return DHTS_E_INVALID;
    const int B = net_block(d), L = d->n_lanes, C = d->n_cells;
    const size_t lds = net_fwd_lds_base(L, C) + net_staged_bytes(L, d->n_inter_sq, d->n_action);
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)net_macro_fwd_kernel<true, false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    net_macro_fwd_kernel<true, false, 1024><<<d->n_replicas, B, lds, (hipStream_t)stream>>>(
        d->n_replicas, L, C, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max, d->static_speed,
        d->vehicle_length, net_tables(t), action, nullptr, nullptr, nullptr, queue, reward, nullptr, err);
    if (hipGetLastError() != hipSuccess) return DHTS_E_LAUNCH;
    if (dhts_opt_reward_chain) return dhts_launch_reward_chain(d->n_replicas, d->n_steps, L, queue, nullptr, 1, d->dt, 0, reward, 1, stream);
    return DHTS_OK;
}

int dhts_net_macro_rollout_bwd(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, const float *hist,
                               const float *tape, const float *kc, const float *queue, const float *g_reward, float *g_action,
                               const float *workspace, dhts_error *err, void *stream) {
    if (!net_desc_ok(d) || !net_tables_ok(t) || !action || !hist || !tape || !kc || !queue || !g_action || !workspace)
        return DHTS_E_INVALID;
    const int B = net_block(d), L = d->n_lanes, C = d->n_cells;
    const int E = t->n_edges > 0 ? t->n_edges : 1;
    const size_t lds = net_bwd_lds_base(L, C, E, d->n_inter_sq) + net_staged_bytes(L, d->n_inter_sq, d->n_action);
    if (lds > 160 * 1024) return DHTS_E_INVALID;
#define DHTS_NET_BWD_LAUNCH(MB)                                                                                                       \
    {                                                                                                                                 \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)net_macro_bwd_kernel<MB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds) != hipSuccess)                                                          \
            return DHTS_E_LAUNCH;                                                                                                     \
        net_macro_bwd_kernel<MB><<<d->n_replicas, B, lds, (hipStream_t)stream>>>(                                                     \
            d->n_replicas, L, C, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max, d->static_speed,       \
            d->vehicle_length, net_tables(t), action, hist, reinterpret_cast<const float4 *>(tape), kc, queue, g_reward, g_action,    \
            workspace, err);                                                                                                          \
    }
    if (B <= 512) DHTS_NET_BWD_LAUNCH(512) else DHTS_NET_BWD_LAUNCH(1024)
#undef DHTS_NET_BWD_LAUNCH
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

}  // extern "C"
