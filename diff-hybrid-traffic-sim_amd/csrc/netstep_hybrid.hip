// netstep_hybrid.hip -- road networks of ANY size and ANY mix of ARZ cell lanes and IDM vehicle lanes, step by step, on gfx950
// (dhts_netstep_rollout_fwd / _bwd, include/dhts.h).
//
// The fused network kernels (hybrid_kernels.hip) keep a replica in ONE workgroup: cells + lanes <= 960, <= 64 IDM lanes, <= 128
// vehicles.  The reference's environment builds any grid in any mode (example/control/itscp/_env.py:221-506); beyond those limits
// an episode used to run lane by lane through the mirror classes (one launch + host work per lane and step: minutes).  Here a step
// is a handful of launches over flat device arrays and all T steps are enqueued by ONE host call:
//   ns_boundary_fwd    blocks >= 1: the two ghost cells of every ARZ lane (ItscpRoadNetwork.setup_macro_boundary, _simulator.py:56-137);
//                      block 0 (the micro side, one workgroup): admission on micro source lanes (:153-174), head gap of every occupied
//                      IDM lane (:139-276 over RoadNetwork.setup_micro_boundary, road_network.py:429-580) as forward-mode duals
//                      w.r.t. (head p, v; leader p, v; three signals), then the IDM step of every vehicle
//                      (MicroLane.forward + dMicroLane._backward, _micro_lane.py:131-214, dmicro_lane.py:87-127)
//   dhts_macro_step_fwd  once per group of ARZ lanes with equal (cells, cell length): the straight-lane operator on the state
//                      arrays in place (cells are stored group-major)
//   ns_convert_fwd     one workgroup: flux capacitors (conversion.py:32-45), the hand-off events in lane-id order by one lane of
//                      wavefront 0 over a candidate bitmap (RoadNetwork.conversion, road_network.py:113-170; conversion.py:11-215)
//                      with an event list, then the queue loss of the committed state: ordered running mean over cells AND
//                      vehicles in lane order (_env.py:586-618, 664-742; rms.py)
// Reverse, newest step first: ns_micro_bwd (loss taps -> events undone in reverse -> speed cotangents folded -> IDM adjoint ->
// head-gap Jacobians to head / leader / signals -> admission undone), dhts_macro_step_bwd per group, ns_ghosts_bwd (+ gather).
// Vehicles live in per-lane slot arrays [micro lane][capacity], slot 0 = tail, slot n - 1 = head (MicroLane.curr_vehicle order,
// _micro_lane.py:31-34); a hand-off moves a vehicle's slot, and the reverse sweep moves the cotangents back the same way, so no
// vehicle ids and no generic tape exist.  Everything that crosses lanes is accumulated in a fixed order (no float atomics):
// results repeat bit for bit.
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"
#include "arz_device.hpp"
#include "idm_device.hpp"
#include "net_device.hpp"

namespace dhts {

constexpr int kNsBlock = 1024;
constexpr long long kNsWindow = 100000;       // RunningMean(100_000), _env.py:122
constexpr int kNsRouteMax = 32;               // MAX_ROUTE_LENGTH, road_network.py:17
enum { NS_EV_SPAWN = 1, NS_EV_CHANGE = 2, NS_EV_DESPAWN = 3, NS_EV_DEPOSIT = 4, NS_EV_DEPCELL = 5, NS_EV_CAPSERIAL = 6 };
enum { NS_CAP_CHARGED = 1, NS_CAP_SERIAL = 2, NS_CAP_OVERWRITTEN = 4 };

struct NsEvent {            // 16 words
    int kind, lane, tgt, cell;
    float overlap, f1, n_r, a, speed, dx, pad0, pad1, pad2, pad3, pad4, pad5;
};
struct NsHeadGap {          // 20 words: d (head_position_delta, head_speed_delta) / d (p_h, v_h, p_l, v_l, s_prev, s_cur, s_next)
    float jp[7], jv[7];
    int leader;             // micro slot of the leader's lane (-1: none)
    int sig[3];             // lane id whose signal is variable 4 + w (-1: a constant)
    int valid, pad;
};
struct NsCounters {
    int n_spawned, n_deposits, n_events, draws_used;
    long long sig_n, loss_n;
    double sig_sum, loss_sum;
    float reward, reward_cut;
    int pad[2];
};

struct NsLayout {           // byte offsets into the workspace
    size_t ghost, g_ghost, slot, own_hist, tape, P, V, A, vroute, vcur, lane_n, rused, capv, counters, sigS, lossS, hg, idm_tape,
        nv_idm, adm, nv_post, vx, kc_cell, kc_veh, caprec, capflag, ev, ev_off, G, gP, gV, gA, g_cap, g_own, g_act, n_bwd, ghd, pF, pAB,
        ptape, total;
    size_t tape_step;       // floats of macro tape per step
};
__host__ __device__ inline size_t ns_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct NsArgs {
    int L, C, T, sq, F, n_action, Lm, cap, ncap, hard, route_stride, n_draws, max_events, loss_steps;
    float um, dtf, vlen, s0f;
    double um_d, dt_d, vlen_d;
    const int32_t *lane_ncell, *lane_off, *sig_kind, *inter, *lane_macro, *lane_gpos, *micro_lanes, *lane_mslot, *cap_lanes, *lane_cslot,
        *lane_source;
    const double *lane_dx, *lane_len;
    const int32_t *left_src, *left_gate, *right_src, *conv_next;
    const double *schedule;
    const int32_t *routes, *route_ptr;
    const double *draws;
    const int32_t *nxt_ptr, *nxt_idx, *prv_ptr, *prv_idx, *inter_ptr, *inter_idx;
    const int32_t *if_lane, *cell_lane;      // the persistent kernels' maps: interface -> lane, cell -> lane (NI = cells + ARZ lanes)
    int NI, n_edges, n_islots, stage;          // stage: the persistent kernels copy the static tables into LDS (they fit)
    // what else a persistent kernel keeps in LDS (null = in the workspace / the history): the state rows t and t + 1 ([2][4][C], row r in
    // copy r & 1), the cotangent planes ([2][3][C]), the per-step table rows of steps t and t + 1 ([2][L] each), ghosts / slots / ghost cotangents
    float *st, *gl, *gh, *sl;
    double *gg;
    int32_t *row_i;        // [2][4][L]: left_src, left_gate, right_src, conv_next
    double *row_d;         // [2][L]: schedule
    long long table_stride, draws_stride;    // elements between replicas in the [T][L] tables / the draws (0 = shared)
    char *ws;
    NsLayout lo;
    float *hist, *queue, *reward;
    int32_t *counts;
    dhts_error *err;
};

template <typename T> __device__ __forceinline__ T *ns_ptr(const NsArgs &a, size_t off) { return reinterpret_cast<T *>(a.ws + off); }
// the state before / after step t, the cotangent planes of the state after step r, the per-step table rows of step t
__device__ __forceinline__ float *ns_state(const NsArgs &a, int r) {
    return a.st ? a.st + (size_t)(r & 1) * 4 * a.C : a.hist + (size_t)r * 4 * a.C;
}
__device__ __forceinline__ float *ns_G(const NsArgs &a, int r) {
    return (a.gl ? a.gl : ns_ptr<float>(a, a.lo.G)) + (size_t)(r & 1) * 3 * a.C;
}
struct NsRow { const int32_t *left_src, *left_gate, *right_src, *conv_next; const double *schedule; };
__device__ __forceinline__ NsRow ns_row(const NsArgs &a, int t) {
    NsRow r;
    if (a.row_i) {
        const int32_t *b = a.row_i + (size_t)(t & 1) * 4 * a.L;
        r.left_src = b; r.left_gate = b + a.L; r.right_src = b + 2 * a.L; r.conv_next = b + 3 * a.L;
        r.schedule = a.row_d + (size_t)(t & 1) * a.L;
    } else {
        const size_t o = (size_t)t * a.L;
        r.left_src = a.left_src + o; r.left_gate = a.left_gate + o; r.right_src = a.right_src + o; r.conv_next = a.conv_next + o;
        r.schedule = a.schedule + o;
    }
    return r;
}
__device__ __forceinline__ float *ns_ghosts(const NsArgs &a) { return a.gh ? a.gh : ns_ptr<float>(a, a.lo.ghost); }
__device__ __forceinline__ float *ns_slots(const NsArgs &a) { return a.sl ? a.sl : ns_ptr<float>(a, a.lo.slot); }
__device__ __forceinline__ double *ns_gghost(const NsArgs &a) { return a.gg ? a.gg : ns_ptr<double>(a, a.lo.g_ghost); }

// inclusive block scan (blockDim.x = multiple of 64, <= 1024); `w` = 16 elements of LDS scratch; returns the inclusive prefix,
// `total` = the block's sum.  Two barriers.
template <typename T>
__device__ __forceinline__ T ns_block_scan(T x, T *w, T &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    T s = wave_scan_add(x);
    if (lane == 63) w[wv] = s;
    __syncthreads();
    T off = 0, tot = 0;
    for (int k = 0; k < nw; ++k) { const T v = w[k]; if (k < wv) off += v; tot += v; }
    __syncthreads();
    total = tot;
    return s + off;
}

// ---- forward-mode duals of the head gap: value + gradient w.r.t. (p_h, v_h, p_l, v_l, s_prev, s_cur, s_next); every component follows
// the float32 rule of the reference's operator (sum, product, quotient, sigmoid), so the VALUES are the reference's bit for bit
struct D7 { float v; float g[7]; };
__device__ __forceinline__ D7 d7_c(float v) { D7 x; x.v = v;
#pragma unroll
    for (int i = 0; i < 7; ++i) x.g[i] = 0.f;
    return x; }
__device__ __forceinline__ D7 d7_var(float v, int i) { D7 x = d7_c(v); x.g[i] = 1.f; return x; }
__device__ __forceinline__ D7 d7_add(D7 a, D7 b) { D7 x; x.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 7; ++i) x.g[i] = a.g[i] + b.g[i];
    return x; }
__device__ __forceinline__ D7 d7_sub(D7 a, D7 b) { D7 x; x.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 7; ++i) x.g[i] = a.g[i] - b.g[i];
    return x; }
__device__ __forceinline__ D7 d7_mul(D7 a, D7 b) { D7 x; x.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 7; ++i) x.g[i] = a.g[i] * b.v + b.g[i] * a.v;
    return x; }
__device__ __forceinline__ D7 d7_div(D7 a, D7 b) { D7 x; x.v = a.v / b.v;
    const float ia = 1.f / b.v, ib = -((a.v / b.v) / b.v);
#pragma unroll
    for (int i = 0; i < 7; ++i) x.g[i] = a.g[i] * ia + b.g[i] * ib;
    return x; }
__device__ __forceinline__ D7 d7_soft(D7 a, float k) {          // dmath.operation.sigmoid(value, constant = k)
    float s, ds;
    soft_switch_both(a.v, k, s, ds);
    D7 x; x.v = s;
#pragma unroll
    for (int i = 0; i < 7; ++i) x.g[i] = a.g[i] * ds;
    return x; }
__device__ __forceinline__ D7 d7_pos_or_zero(D7 a) { return a.v > 0.f ? a : d7_c(0.f); }     // x if x > 0 else 0.0

__device__ __forceinline__ float ns_lane_signal(const NsArgs &a, const float *action, int t, int lid, bool hard, float *ds_da, int *a_index) {
    const int kd = a.sig_kind[lid];
    if (kd == 0) { if (ds_da) *ds_da = 0.f; if (a_index) *a_index = -1; return 1.f; }
    float we, ns, av, pr; int ai;
    phase_signal_at(action, a.n_action, a.sq, a.F, t / a.F, t % a.F, a.inter[lid], we, ns, av, pr, ai, hard);
    if (ds_da) {
        const float z = (av - pr) * kSigK;
        const bool sat = hard || z < -16.f || z > 16.f;
        *ds_da = sat ? 0.f : (kd == 1 ? we * (1.f - we) * kSigK : -(ns * (1.f - ns) * kSigK));
    }
    if (a_index) *a_index = ai;
    return kd == 1 ? we : ns;
}

// shift the first n slots of a micro lane's rows one slot towards the head (a vehicle enters at the tail, slot 0)
__device__ __forceinline__ void ns_shift_in(float *P, float *V, float *A, int *vr, int *vc, size_t base, int n) {
    for (int i = n; i > 0; --i) {
        P[base + i] = P[base + i - 1]; V[base + i] = V[base + i - 1]; A[base + i] = A[base + i - 1];
        vr[base + i] = vr[base + i - 1]; vc[base + i] = vc[base + i - 1];
    }
}

// =====================================================================================================================
// forward, part 1: boundaries of step t from the state before it; IDM steps
// =====================================================================================================================
// ghost (lane, side) = item j of every ARZ lane: _simulator.py:56-137, road_network.py:299-362
__device__ __forceinline__ void ns_ghost_fwd_item(const NsArgs &a, int t, const float *__restrict__ action, int j) {
    const int L = a.L, C = a.C;
    const bool hard = a.hard != 0;
    const float um = a.um;
    const float *cur = ns_state(a, t);
    const NsRow rw = ns_row(a, t);
    {
        if (j >= 2 * L) return;
        const int lane = j >> 1, side = j & 1;
        const float *own_in = ns_ptr<float>(a, a.lo.own_hist) + (size_t)t * 2 * L;
        float *own_out = ns_ptr<float>(a, a.lo.own_hist) + (size_t)(t + 1) * 2 * L;
        if (!a.lane_macro[lane]) {
            if (side == 1) { own_out[2 * lane] = own_in[2 * lane]; own_out[2 * lane + 1] = own_in[2 * lane + 1]; }
            return;
        }
        float fr, fu, fy, fq;
        if (side == 0) {
            const int ls = rw.left_src[lane], lg = rw.left_gate[lane];
            if (ls == -1) {                    // source lane: Python floats in the reference (_simulator.py:68-71)
                const double sched = rw.schedule[lane];
                const double gu = a.um_d * (1. - sqrt(fmax(sched, 0.) + kEps));
                fr = (float)sched; fu = (float)gu; fy = 0.f; fq = (float)gu;          // y = r (u - u_eq(r)) = 0
            } else {
                float gr = 0.f, gu = um;       // ls == -3: the lane's own stored upstream ghost (its single upstream lane is micro)
                if (ls >= 0) { const int last = a.lane_off[ls] + a.lane_ncell[ls] - 1; gr = cur[last]; gu = cur[2 * C + last]; }
                const float s = lg == -1 ? 0.f : (lg == -2 ? 1.f : ns_lane_signal(a, action, t, lg, hard, nullptr, nullptr));
                fr = gr * s + 0.f * (1.0f - s);
                fu = gu * s + um * (1.0f - s);
                glue_from_r_u(fr, fu, um, fy, fq);
            }
        } else {
            const int rs = rw.right_src[lane];
            float gr = own_in[2 * lane], gu = own_in[2 * lane + 1];
            if (rs >= 0) { const int first = a.lane_off[rs]; gr = cur[first]; gu = cur[2 * C + first]; }
            const float sg = ns_lane_signal(a, action, t, lane, hard, nullptr, nullptr);
            const float s2 = hard ? (sg > 0.5f ? 1.f : 0.f) : soft_switch(sg - 0.5f, kSigK);
            fr = s2 * gr + (1.0f - s2) * 1.0f;
            fu = s2 * gu + (1.0f - s2) * 0.0f;
            glue_from_r_u(fr, fu, um, fy, fq);
            own_out[2 * lane] = fr; own_out[2 * lane + 1] = fu;
        }
        float *g = ns_ghosts(a) + ((size_t)a.lane_gpos[lane] * 2 + side) * 4;
        g[0] = fr; g[1] = fy; g[2] = fu; g[3] = fq;
    }
}

// the micro side of a step's boundary + the IDM steps: ONE workgroup (all its threads call; hd_s = [Lm][2] floats of LDS)
__device__ __forceinline__ void ns_micro_fwd(const NsArgs &a, int t, const float *__restrict__ action, float *hd_s) {
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a.L;
    const bool hard = a.hard != 0;
    const size_t row = (size_t)t * L;
    const int Lm = a.Lm, cap = a.cap;
    if (Lm == 0) return;
    __shared__ double scan_d[16];
    __shared__ int scan_i[16];
    const size_t plane = (size_t)Lm * cap;
    float *P0 = ns_ptr<float>(a, a.lo.P) + (size_t)(t & 1) * plane, *P1 = ns_ptr<float>(a, a.lo.P) + (size_t)((t + 1) & 1) * plane;
    float *V0 = ns_ptr<float>(a, a.lo.V) + (size_t)(t & 1) * plane, *V1 = ns_ptr<float>(a, a.lo.V) + (size_t)((t + 1) & 1) * plane;
    float *A_ = ns_ptr<float>(a, a.lo.A);
    int *vroute = ns_ptr<int>(a, a.lo.vroute), *vcur = ns_ptr<int>(a, a.lo.vcur), *lane_n = ns_ptr<int>(a, a.lo.lane_n);
    int *rused = ns_ptr<int>(a, a.lo.rused);
    NsCounters *cnt = ns_ptr<NsCounters>(a, a.lo.counters);
    int *adm = hard ? nullptr : ns_ptr<int>(a, a.lo.adm) + (size_t)t * Lm;
    const float vlen = a.vlen;

    // ---- admission on micro source lanes (_simulator.py:153-174): room for half a vehicle at the entrance, then ONE draw of the
    // host's stream per such lane, lanes in id order ----
    if (a.lane_source != nullptr) {
        int carry = 0;
        const int used0 = cnt->draws_used;
        for (int base = 0; base < Lm; base += B) {
            const int m = base + tid;
            int has = 0, l = -1, n = 0;
            if (m < Lm) {
                l = a.micro_lanes[m];
                if (a.lane_source[l]) {
                    n = lane_n[m];
                    const float room = n ? P0[(size_t)m * cap] - 0.5f * vlen : (float)a.lane_len[l];
                    has = room > vlen * 0.5f;
                }
            }
            int total;
            const int rank = ns_block_scan<int>(has, scan_i, total) - has;
            int admitted = 0;
            if (has) {
                const int idx = used0 + carry + rank;
                if (idx >= a.n_draws) net_fault(a.err, DHTS_FAULT_CAPACITY, t, l, -1);
                else {
                    const double draw = a.draws[idx];
                    const int r_lo = a.route_ptr[l], r_n = a.route_ptr[l + 1] - r_lo;
                    if (draw < ns_row(a, t).schedule[l] && rused[l] < r_n) {
                        if (n >= cap) net_fault(a.err, DHTS_FAULT_CAPACITY, t, l, n);
                        else {
                            const size_t b = (size_t)m * cap;
                            ns_shift_in(P0, V0, A_, vroute, vcur, b, n);
                            P0[b] = 0.f; V0[b] = 0.f; A_[b] = vlen; vroute[b] = r_lo + rused[l]; vcur[b] = 0;
                            rused[l] += 1; lane_n[m] = n + 1;
                            admitted = 1;
                        }
                    }
                }
            }
            if (adm && m < Lm) adm[m] = admitted;
            int tot_adm;
            (void)ns_block_scan<int>(admitted, scan_i, tot_adm);
            if (tid == 0 && tot_adm) cnt->n_spawned += tot_adm;
            carry += total;
        }
        __syncthreads();
        if (tid == 0) cnt->draws_used = used0 + carry;
    } else if (adm) {
        for (int m = tid; m < Lm; m += B) adm[m] = 0;
    }
    __syncthreads();

    // ---- head gaps (lane-id order for the running mean of the final signal) ----
    NsHeadGap *hgrow = hard ? nullptr : ns_ptr<NsHeadGap>(a, a.lo.hg) + (size_t)t * Lm;
    double *sigS = ns_ptr<double>(a, a.lo.sigS);
    long long sig_n0 = cnt->sig_n;
    double sig_sum0 = cnt->sig_sum;
    __syncthreads();
    for (int base = 0; base < Lm; base += B) {
        const int m = base + tid;
        bool occ = false;
        D7 green_dp = d7_c(1000.f), green_dv = d7_c(0.f), red_dp = d7_c(0.f), fin = d7_c(0.f);
        int leader = -1, sgl[3] = {-1, -1, -1};
        if (m < Lm) {
            const int n = lane_n[m];
            if (n > 0) {
                occ = true;
                const int l = a.micro_lanes[m];
                const size_t hs = (size_t)m * cap + n - 1;
                const D7 hp = d7_var(P0[hs], 0), hv = d7_var(V0[hs], 1);
                const int *route = a.routes + (size_t)vroute[hs] * a.route_stride;
                const int cursor = vcur[hs];
                int rlen = 0;
                while (rlen < a.route_stride && route[rlen] >= 0) rlen++;
                const D7 Lc = d7_c((float)a.lane_len[l]);
                // leader further along the route (road_network.py:459-580)
                D7 reach = d7_sub(d7_sub(Lc, hp), d7_c(vlen * 0.5f));
                for (int k = cursor; k < rlen - 1; k++) {
                    const int there = route[k + 1];
                    if (a.lane_macro[there]) break;
                    const int ms = a.lane_mslot[there];
                    if (lane_n[ms]) {
                        const size_t ts = (size_t)ms * cap;
                        const D7 lp = d7_var(P0[ts], 2), lv = d7_var(V0[ts], 3);
                        const D7 gap = d7_add(reach, d7_sub(lp, d7_c(vlen * 0.5f)));
                        green_dp = d7_pos_or_zero(gap);
                        green_dv = d7_sub(hv, lv);
                        leader = ms;
                        break;
                    }
                    reach = d7_add(reach, d7_c((float)a.lane_len[there]));
                }
                // red light: stop line at the lane end (_simulator.py:188-194)
                red_dp = d7_pos_or_zero(d7_sub(d7_sub(Lc, hp), d7_c(vlen * 0.5f)));
                const bool prev_exist = cursor > 0, next_exist = cursor < rlen - 1;
                D7 prev_s = d7_c(0.f), next_s = d7_c(0.f);
                if (prev_exist && !hard) prev_s = d7_soft(d7_sub(d7_c(0.f), hp), 16.f);
                const D7 curr_s = hard ? d7_c(1.f) : d7_mul(d7_soft(hp, 16.f), d7_soft(d7_sub(Lc, hp), 16.f));
                if (next_exist && !hard) next_s = d7_soft(d7_sub(hp, Lc), 16.f);
                const D7 total = d7_add(d7_add(prev_s, curr_s), next_s);
                for (int w = 0; w < 3; w++) {
                    if ((w == 0 && !prev_exist) || (w == 2 && !next_exist)) continue;
                    const int lid = route[cursor + w - 1];
                    const D7 sc = w == 0 ? prev_s : (w == 1 ? curr_s : next_s);
                    D7 sv;
                    if (a.sig_kind[lid] == 0) sv = d7_c(1.f);
                    else { sv = d7_var(ns_lane_signal(a, action, t, lid, hard, nullptr, nullptr), 4 + w); sgl[w] = lid; }
                    fin = d7_add(fin, d7_mul(d7_div(sc, total), sv));
                }
            }
        }
        // ordered running mean of the final signal over the occupied lanes (signal_rms: _simulator.py:234-262; rms.py)
        float hdp = 1000.f, hdv = 0.f;
        if (hard) {
            if (occ) { const bool green = fin.v >= 0.5f; hdp = green ? green_dp.v : red_dp.v; hdv = green ? green_dv.v : 0.f; }
        } else {
            int tot_n; double tot_s;
            const int rank = ns_block_scan<int>(occ ? 1 : 0, scan_i, tot_n);
            const double incl = ns_block_scan<double>(occ ? (double)fin.v : 0., scan_d, tot_s);
            if (occ) {
                const long long i = sig_n0 + rank - 1;               // 0-based index of this sample in the stream
                const double S = sig_sum0 + incl;
                sigS[i] = S;
                const double mean = (i + 1 > kNsWindow) ? (S - sigS[i - kNsWindow]) / (double)kNsWindow : S / (double)(i + 1);
                const float k2 = 32.f / fabsf((float)mean);
                const D7 fs = d7_soft(d7_sub(fin, d7_c(0.5f)), k2);
                const D7 one_m = d7_sub(d7_c(1.f), fs);
                const D7 o_dp = d7_add(d7_mul(green_dp, fs), d7_mul(red_dp, one_m));
                const D7 o_dv = d7_add(d7_mul(green_dv, fs), d7_mul(d7_c(0.f), one_m));
                hdp = o_dp.v; hdv = o_dv.v;
                NsHeadGap h;
#pragma unroll
                for (int q = 0; q < 7; ++q) { h.jp[q] = o_dp.g[q]; h.jv[q] = o_dv.g[q]; }
                h.leader = leader; h.sig[0] = sgl[0]; h.sig[1] = sgl[1]; h.sig[2] = sgl[2]; h.valid = 1; h.pad = 0;
                hgrow[m] = h;
            } else if (m < Lm) {
                hgrow[m].valid = 0;
            }
            sig_n0 += tot_n; sig_sum0 += tot_s;
        }
        if (m < Lm) { hd_s[2 * m] = hdp; hd_s[2 * m + 1] = hdv; }
    }
    __syncthreads();
    if (tid == 0 && !hard) { cnt->sig_n = sig_n0; cnt->sig_sum = sig_sum0; }

    // ---- one IDM step of every vehicle (explicit Euler; _micro_lane.py:131-214, _idm.py:6-50, didm.py:13-103) ----
    IdmParams prm;              // MicroVehicle.default_micro_vehicle(speed_limit), micro_vehicle.py:31-72
    prm.a_max = a.um_d * 1.0; prm.a_pref = a.um_d * 0.8; prm.v_target = a.um_d * 0.9; prm.min_space = a.vlen_d * 0.1; prm.time_pref = 0.1;
    prm.length = a.vlen_d;
    float4 *tape = hard ? nullptr : ns_ptr<float4>(a, a.lo.idm_tape) + (size_t)t * plane;
    int *nv_idm = hard ? nullptr : ns_ptr<int>(a, a.lo.nv_idm) + (size_t)t * Lm;
    for (size_t idx = tid; idx < plane; idx += B) {
        const int m = (int)(idx / cap), i = (int)(idx - (size_t)m * cap);
        const int n = lane_n[m];
        if (i == 0 && nv_idm) nv_idm[m] = n;
        if (i >= n) continue;
        const double p = P0[idx], v = V0[idx];
        double dp, dv;
        if (i == n - 1) { dp = (double)hd_s[2 * m]; dv = (double)hd_s[2 * m + 1]; }
        else { dp = fabs((double)P0[idx + 1] - p) - ((prm.length + prm.length) * 0.5); dv = v - (double)V0[idx + 1]; }
        IdmStep o;
        idm_step_ieee(p, v, dp, dv, prm, a.dt_d, o);
        if (o.collided) net_fault(a.err, DHTS_FAULT_COLLISION, t, a.micro_lanes[m], i);
        P1[idx] = o.np; V1[idx] = o.nv;
        if (tape) tape[idx] = make_float4(o.dE[2], o.dE[3], o.dLd[2], o.dLd[3]);
    }
}

__global__ void __launch_bounds__(kNsBlock) ns_boundary_fwd_kernel(NsArgs a, int t, const float *__restrict__ action) {
    extern __shared__ float hd_dyn[];                 // [Lm][2] head gaps of this step
    if (blockIdx.x > 0) ns_ghost_fwd_item(a, t, action, (blockIdx.x - 1) * blockDim.x + threadIdx.x);
    else ns_micro_fwd(a, t, action, hd_dyn);
}

// =====================================================================================================================
// forward, part 2: hand-offs in lane-id order on the committed state, then the queue loss
// =====================================================================================================================
__device__ __forceinline__ void ns_push_event(const NsArgs &a, NsCounters *cnt, int t, const NsEvent &e) {
    if (a.hard) return;
    const int k = cnt->n_events;
    if (k >= a.max_events) { net_fault(a.err, DHTS_FAULT_CAPACITY, t, e.lane, -2); return; }
    ns_ptr<NsEvent>(a, a.lo.ev)[k] = e;
    cnt->n_events = k + 1;
}

__device__ __forceinline__ void ns_convert(const NsArgs &a, int t, int *cand) {      // cand: [L] ints of LDS (candidate flags of the event walk)
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a.L, C = a.C, Lm = a.Lm, cap = a.cap;
    const bool hard = a.hard != 0;
    const float um = a.um, vlen = a.vlen, dtf = a.dtf, s0f = a.s0f;
    __shared__ double scan_d[16];
    __shared__ int scan_i[16];
    float *nxt = ns_state(a, t + 1);                   // committed state: the lanes' steps wrote it
    const NsRow rw = ns_row(a, t);
    float *Rn = nxt, *Yn = nxt + C, *Un = nxt + 2 * C;
    const size_t row = (size_t)t * L;
    const size_t plane = (size_t)Lm * cap;
    float *P1 = Lm ? ns_ptr<float>(a, a.lo.P) + (size_t)((t + 1) & 1) * plane : nullptr;
    float *V1 = Lm ? ns_ptr<float>(a, a.lo.V) + (size_t)((t + 1) & 1) * plane : nullptr;
    float *A_ = ns_ptr<float>(a, a.lo.A);
    int *vroute = ns_ptr<int>(a, a.lo.vroute), *vcur = ns_ptr<int>(a, a.lo.vcur), *lane_n = ns_ptr<int>(a, a.lo.lane_n);
    int *rused = ns_ptr<int>(a, a.lo.rused);
    float *capv = ns_ptr<float>(a, a.lo.capv);
    NsCounters *cnt = ns_ptr<NsCounters>(a, a.lo.counters);
    float4 *caprec = ns_ptr<float4>(a, a.lo.caprec) + (hard ? 0 : (size_t)t * a.ncap);
    int *capflag = ns_ptr<int>(a, a.lo.capflag) + (hard ? 0 : (size_t)t * a.ncap);
    int *ev_off = ns_ptr<int>(a, a.lo.ev_off);

    if (Lm > 0) {
        for (int l = tid; l < L; l += B) cand[l] = 0;
        __syncthreads();
        // ---- flux capacitors: += r u dt of the last cell (conversion.py:32-45) ----
        for (int k = tid; k < a.ncap; k += B) {
            const int l = a.cap_lanes[k];
            const int m = rw.conv_next[l];
            int flag = 0;
            if (m >= 0 && !a.lane_macro[m]) {
                const int last = a.lane_off[l] + a.lane_ncell[l] - 1;
                const float rl = Rn[last], ul = Un[last];
                const float level_old = capv[k];
                const float level = level_old + (rl * ul) * dtf;
                capv[k] = level;
                caprec[k] = make_float4(rl, Yn[last], ul, level_old);
                flag = NS_CAP_CHARGED;
                cand[l] = level >= vlen;
            }
            capflag[k] = flag;
        }
        for (int m = tid; m < Lm; m += B) {
            const int n = lane_n[m];
            if (n > 0) { const int l = a.micro_lanes[m]; cand[l] = P1[(size_t)m * cap + n - 1] >= (float)a.lane_len[l]; }
        }
        __syncthreads();
        if (tid == 0 && !hard) ev_off[t] = cnt->n_events;
        // ---- the hand-off events, serially in lane-id order (RoadNetwork.conversion, road_network.py:113-170): wavefront 0 finds the
        // next candidate with a ballot, its lane 0 acts; an event can make a LATER lane a candidate ----
        if (tid < 64) {
            for (int base = 0; base < L; base += 64) {
                int done_below = 0;            // bits [0, done_below) of this chunk are dealt with
                while (true) {
                    const int l_ = base + tid;
                    const bool c = l_ < L && tid >= done_below && cand[l_] != 0;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(c);
                    if (!mask) break;
                    const int bit = __builtin_ctzll(mask);
                    const int l = base + bit;
                    done_below = bit + 1;
                    if (tid == 0) {
                        if (a.lane_macro[l]) {
                            // macro -> micro (conversion.py:16-73)
                            const int k = a.lane_cslot[l];
                            const int m = rw.conv_next[l];
                            if (capflag[k] & NS_CAP_SERIAL) { NsEvent e = {}; e.kind = NS_EV_CAPSERIAL; e.lane = l; ns_push_event(a, cnt, t, e); }
                            const float level = capv[k];
                            const int ms = a.lane_mslot[m];
                            const int n = lane_n[ms];
                            const size_t b = (size_t)ms * cap;
                            const float space = n ? P1[b] - 0.5f * vlen : (float)a.lane_len[m];
                            if (level >= vlen && space >= vlen * 1.0f) {
                                const int r_lo = a.route_ptr[m], r_n = a.route_ptr[m + 1] - r_lo;
                                if (r_n <= 0 || n >= cap) net_fault(a.err, DHTS_FAULT_CAPACITY, t, m, n);
                                else {
                                    const int rrow = r_lo + rused[m] % r_n;
                                    rused[m] += 1;
                                    ns_shift_in(P1, V1, A_, vroute, vcur, b, n);
                                    P1[b] = 0.f; V1[b] = caprec[k].z;
                                    A_[b] = level - (float)((double)level - (double)vlen);
                                    vroute[b] = rrow; vcur[b] = 0;
                                    lane_n[ms] = n + 1;
                                    capv[k] = (float)((double)level - (double)vlen);
                                    cnt->n_spawned += 1;
                                    NsEvent e = {}; e.kind = NS_EV_SPAWN; e.lane = l; e.tgt = m;
                                    ns_push_event(a, cnt, t, e);
                                }
                            }
                        } else {
                            const int ms = a.lane_mslot[l];
                            const int n = lane_n[ms];
                            if (n > 0) {
                                const size_t hs = (size_t)ms * cap + n - 1;
                                const int *route = a.routes + (size_t)vroute[hs] * a.route_stride;
                                const int cursor = vcur[hs];
                                int rlen = 0;
                                while (rlen < a.route_stride && route[rlen] >= 0) rlen++;
                                const int nid = cursor < rlen - 1 ? route[cursor + 1] : -1;
                                const float Lf = (float)a.lane_len[l];
                                const float hp = P1[hs], hv = V1[hs], ha = A_[hs];
                                if (nid == -1) {                                    // micro -> none (conversion.py:203-215)
                                    if (hp >= Lf) {
                                        lane_n[ms] = n - 1;
                                        NsEvent e = {}; e.kind = NS_EV_DESPAWN; e.lane = l; ns_push_event(a, cnt, t, e);
                                    }
                                } else if (!a.lane_macro[nid]) {                    // micro -> micro (:175-200)
                                    if (hp >= Lf) {
                                        const int m2 = a.lane_mslot[nid];
                                        const int n2 = lane_n[m2];
                                        if (n2 >= cap) net_fault(a.err, DHTS_FAULT_CAPACITY, t, nid, n2);
                                        else {
                                            const int vr_ = vroute[hs];
                                            lane_n[ms] = n - 1;
                                            const size_t b2 = (size_t)m2 * cap;
                                            ns_shift_in(P1, V1, A_, vroute, vcur, b2, n2);
                                            const float np_ = hp - Lf;
                                            P1[b2] = np_; V1[b2] = hv; A_[b2] = ha; vroute[b2] = vr_; vcur[b2] = cursor + 1;
                                            lane_n[m2] = n2 + 1;
                                            if (n2 == 0 && nid > l) cand[nid] = np_ >= (float)a.lane_len[nid];
                                            NsEvent e = {}; e.kind = NS_EV_CHANGE; e.lane = l; e.tgt = nid; ns_push_event(a, cnt, t, e);
                                        }
                                    }
                                } else if (hp > Lf + 1.0f * vlen) {                  // micro -> macro (:76-171)
                                    lane_n[ms] = n - 1;
                                    cnt->n_deposits += 1;
                                    const float front = hp - Lf, rear = front - vlen;
                                    const double dxd = a.lane_dx[nid];
                                    const float dx = (float)dxd;
                                    for (int ci = 0; ci < a.lane_ncell[nid]; ci++) {
                                        const double c_lo = dxd * ci, c_hi = dxd * (ci + 1);
                                        if (!(c_hi > (double)rear && c_lo < (double)front)) break;
                                        const int hi_is_front = (double)front > c_hi, lo_is_rear = (double)rear < c_lo;
                                        const float hi = hi_is_front ? front : (float)c_hi, lo = lo_is_rear ? rear : (float)c_lo;
                                        const float overlap = dx + vlen - (hi - lo);
                                        const int cell = a.lane_off[nid] + ci;
                                        float n_r = Rn[cell] + (ha / vlen) * (overlap / dx);
                                        if (n_r > 1.0f - 1e-5f) n_r = n_r - (float)((double)n_r - (1.0 - 1e-5));
                                        else if (n_r < 1e-5f) n_r = n_r - (float)((double)n_r - 1e-5);
                                        NsEvent e = {}; e.kind = NS_EV_DEPCELL; e.lane = l; e.tgt = nid; e.cell = cell;
                                        e.overlap = overlap; e.f1 = (float)(-hi_is_front + lo_is_rear); e.n_r = n_r; e.a = ha; e.speed = hv; e.dx = dx;
                                        ns_push_event(a, cnt, t, e);
                                        Rn[cell] = n_r;
                                        Un[cell] = hv;
                                        float yy, qq;
                                        glue_from_r_u(n_r, hv, um, yy, qq);
                                        Yn[cell] = yy;                          // u_eq keeps its value from before the deposit
                                        // a flux capacitor reading this very cell: one of a LATER lane charges again from the deposited
                                        // state (the reference walks the lanes in id order), one of an EARLIER lane keeps what it read
                                        if (ci == a.lane_ncell[nid] - 1) {
                                            const int k2 = a.ncap ? a.lane_cslot[nid] : -1;
                                            if (k2 >= 0 && (capflag[k2] & NS_CAP_CHARGED)) {
                                                if (nid > l) {
                                                    const float lo_ = caprec[k2].w;
                                                    const float lv_ = lo_ + (n_r * hv) * dtf;
                                                    capv[k2] = lv_;
                                                    caprec[k2] = make_float4(n_r, yy, hv, lo_);
                                                    capflag[k2] |= NS_CAP_SERIAL;
                                                    cand[nid] = 1;               // visited in order: its charge is part of the event list
                                                } else capflag[k2] |= NS_CAP_OVERWRITTEN;
                                            }
                                        }
                                    }
                                    NsEvent e = {}; e.kind = NS_EV_DEPOSIT; e.lane = l; e.tgt = nid; ns_push_event(a, cnt, t, e);
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0 && !hard) ev_off[t + 1] = cnt->n_events;

    // ---- queue-length loss of the committed state (lane-id order; _env.py:664-742 with :586-618) ----
    int *nv_post = (Lm && !hard) ? ns_ptr<int>(a, a.lo.nv_post) + (size_t)t * Lm : nullptr;
    float *vx = (Lm && !hard) ? ns_ptr<float>(a, a.lo.vx) + (size_t)t * plane : nullptr;
    float *kc_cell = hard ? nullptr : ns_ptr<float>(a, a.lo.kc_cell) + (size_t)t * C;
    float *kc_veh = (Lm && !hard) ? ns_ptr<float>(a, a.lo.kc_veh) + (size_t)t * plane : nullptr;
    double *lossS = ns_ptr<double>(a, a.lo.lossS);
    long long n0 = cnt->loss_n;
    double S0 = cnt->loss_sum;
    __syncthreads();
    for (int base = 0; base < L; base += B) {
        const int l = base + tid;
        int cntl = 0, ms = -1, off = 0;
        bool macro = false;
        double sum = 0.;
        if (l < L) {
            macro = a.lane_macro[l] != 0;
            if (macro) {
                cntl = a.lane_ncell[l]; off = a.lane_off[l];
                if (!hard) for (int i = 0; i < cntl; i++) sum += (double)(s0f - Un[off + i]);
            } else {
                ms = a.lane_mslot[l];
                cntl = lane_n[ms];
                if (nv_post) nv_post[ms] = cntl;
                if (!hard) for (int i = 0; i < cntl; i++) sum += (double)(s0f - V1[(size_t)ms * cap + i]);
            }
        }
        float qlen = 0.f;
        if (hard) {
            if (l < L) {
                if (macro) { const float w = (float)a.lane_dx[l]; for (int i = 0; i < cntl; i++) qlen = qlen + (Un[off + i] < s0f ? 1.f : 0.f) * (Rn[off + i] * w / vlen); }
                else for (int i = 0; i < cntl; i++) qlen = qlen + (V1[(size_t)ms * cap + i] < s0f ? 1.f : 0.f);
            }
        } else {
            int tot_n; double tot_s;
            const int n_incl = ns_block_scan<int>(cntl, scan_i, tot_n);
            const double s_incl = ns_block_scan<double>(sum, scan_d, tot_s);
            if (l < L) {
                long long i_g = n0 + (n_incl - cntl);
                double S = S0 + (s_incl - sum);
                const float w = macro ? (float)a.lane_dx[l] : 0.f;
                for (int i = 0; i < cntl; i++, i_g++) {
                    const float x = s0f - (macro ? Un[off + i] : V1[(size_t)ms * cap + i]);
                    S += (double)x;
                    lossS[i_g] = S;
                    const double mean = (i_g + 1 > kNsWindow) ? (S - lossS[i_g - kNsWindow]) / (double)kNsWindow : S / (double)(i_g + 1);
                    const float k = 16.f / fabsf((float)mean);
                    if (macro) { kc_cell[off + i] = k; qlen = qlen + soft_switch(x, k) * (Rn[off + i] * w / vlen); }
                    else { kc_veh[(size_t)ms * cap + i] = k; vx[(size_t)ms * cap + i] = x; qlen = qlen + soft_switch(x, k); }
                }
            }
            n0 += tot_n; S0 += tot_s;
        }
        if (l < L) a.queue[row + l] = (qlen * qlen) * dtf;
    }
    __syncthreads();
    if (tid == 0 && !hard) { cnt->loss_n = n0; cnt->loss_sum = S0; }
}

__global__ void __launch_bounds__(kNsBlock) ns_convert_fwd_kernel(NsArgs a, int t) {
    extern __shared__ int cand_dyn[];
    ns_convert(a, t, cand_dyn);
}

// reward = - sum of the queue terms, lanes outermost (ItscpEnv._reward, _env.py:770-797): one thread per lane sums its steps,
// lane 0 of the block the lanes; counts out
__device__ __forceinline__ void ns_reward(const NsArgs &a, float *part) {          // part: [2][L] floats of LDS
    const int tid = threadIdx.x, B = blockDim.x, L = a.L, T = a.T;
    const int cut = (a.loss_steps > 0 && a.loss_steps < T) ? a.loss_steps : T;
    for (int l = tid; l < L; l += B) {
        float s = 0.f, sc = 0.f;
        for (int t = 0; t < T; t++) { const float q = (-1.0f) * a.queue[(size_t)t * L + l]; s = s + q; if (t < cut) sc = sc + q; }
        part[l] = s; part[L + l] = sc;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, sc = 0.f;
        for (int l = 0; l < L; l++) { s = s + part[l]; sc = sc + part[L + l]; }
        a.reward[0] = s; a.reward[1] = sc;
        const NsCounters *cnt = ns_ptr<NsCounters>(a, a.lo.counters);
        a.counts[0] = cnt->n_spawned; a.counts[1] = cnt->n_deposits; a.counts[2] = cnt->n_events; a.counts[3] = cnt->draws_used;
    }
}
__global__ void __launch_bounds__(kNsBlock) ns_reward_kernel(NsArgs a) {
    extern __shared__ float part_dyn[];
    ns_reward(a, part_dyn);
}

// =====================================================================================================================
// reverse, part 1 (one workgroup): loss taps of the state after step t, the step's events undone newest first, speed cotangents
// folded into (r, y), IDM adjoint, head-gap Jacobians, admission undone
// =====================================================================================================================
__device__ __forceinline__ void ns_shift_out(double *gP, double *gV, double *gA, size_t base, int n) {       // slot 0 leaves
    for (int i = 0; i + 1 < n; ++i) { gP[base + i] = gP[base + i + 1]; gV[base + i] = gV[base + i + 1]; gA[base + i] = gA[base + i + 1]; }
}

__device__ __forceinline__ void ns_micro_bwd(const NsArgs &a, int t, const float *__restrict__ action, const float *__restrict__ g_reward,
                                             double *lds_d) {      // lds_d: [2 Lm + 8 blockDim] doubles of LDS (head-gap cotangents | compact lists)
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a.L, C = a.C, Lm = a.Lm, cap = a.cap;
    const float um = a.um, vlen = a.vlen, dtf = a.dtf, s0f = a.s0f;
    __shared__ int scan_i[16];
    const float *nxt = ns_state(a, t + 1);
    const float *Rn = nxt, *Yn = nxt + C, *Un = nxt + 2 * C;
    float *G = ns_G(a, t + 1);                         // cotangent of (r, y, u) of the state after step t
    float *Gr = G, *Gy = G + C, *Gu = G + 2 * C;
    const size_t row = (size_t)t * L;
    const size_t plane = (size_t)Lm * cap;
    const float grew = g_reward ? g_reward[0] : 1.f;
    const bool taps = a.loss_steps <= 0 || t < a.loss_steps;
    double *gPn = ns_ptr<double>(a, a.lo.gP) + (size_t)((t + 1) & 1) * plane, *gPp = ns_ptr<double>(a, a.lo.gP) + (size_t)(t & 1) * plane;
    double *gVn = ns_ptr<double>(a, a.lo.gV) + (size_t)((t + 1) & 1) * plane, *gVp = ns_ptr<double>(a, a.lo.gV) + (size_t)(t & 1) * plane;
    double *gA = ns_ptr<double>(a, a.lo.gA);
    double *g_cap = ns_ptr<double>(a, a.lo.g_cap);
    int *n_bwd = ns_ptr<int>(a, a.lo.n_bwd);
    const float4 *caprec = ns_ptr<float4>(a, a.lo.caprec) + (size_t)t * a.ncap;
    const int *capflag = ns_ptr<int>(a, a.lo.capflag) + (size_t)t * a.ncap;

    // ---- (a) loss taps: reward = - sum_l q_l^2 dt ----
    if (taps) {
        const float *kc_cell = ns_ptr<float>(a, a.lo.kc_cell) + (size_t)t * C;
        for (int l = tid; l < L; l += B) {
            if (a.lane_macro[l]) {
                const int n = a.lane_ncell[l], off = a.lane_off[l];
                const float w = (float)a.lane_dx[l];
                float qlen = 0.f;
                for (int i = 0; i < n; i++) qlen += soft_switch(s0f - Un[off + i], kc_cell[off + i]) * (Rn[off + i] * w / vlen);
                const float g_q = (-1.0f * dtf * 2.f * qlen) * grew;
                for (int i = 0; i < n; i++) {
                    float s, ds;
                    soft_switch_both(s0f - Un[off + i], kc_cell[off + i], s, ds);
                    Gr[off + i] += g_q * s * (w / vlen);
                    Gu[off + i] += g_q * (Rn[off + i] * w / vlen) * (-ds);
                }
            } else {
                const int ms = a.lane_mslot[l];
                const int n = ns_ptr<int>(a, a.lo.nv_post)[(size_t)t * Lm + ms];
                const float *vx = ns_ptr<float>(a, a.lo.vx) + (size_t)t * plane + (size_t)ms * cap;
                const float *kv = ns_ptr<float>(a, a.lo.kc_veh) + (size_t)t * plane + (size_t)ms * cap;
                float qlen = 0.f;
                for (int i = 0; i < n; i++) qlen = qlen + soft_switch(vx[i], kv[i]);
                const double seed = (double)((-1.0f * dtf * 2.f * qlen) * grew);
                for (int i = 0; i < n; i++) gVn[(size_t)ms * cap + i] += seed * (double)soft_switch_grad(vx[i], kv[i]) * -1.0;
            }
        }
    }
    __syncthreads();

    if (Lm > 0) {
        // ---- (b) the step's hand-off events, newest first (one thread: they are few) ----
        if (tid == 0) {
            const int *ev_off = ns_ptr<int>(a, a.lo.ev_off);
            const NsEvent *ev = ns_ptr<NsEvent>(a, a.lo.ev);
            for (int k = ev_off[t + 1] - 1; k >= ev_off[t]; --k) {
                const NsEvent e = ev[k];
                if (e.kind == NS_EV_SPAWN) {
                    const int ms = a.lane_mslot[e.tgt];
                    const size_t b = (size_t)ms * cap;
                    const double gv = gVn[b], ga = gA[b];
                    ns_shift_out(gPn, gVn, gA, b, n_bwd[ms]);
                    n_bwd[ms] -= 1;
                    const int kc = a.lane_cslot[e.lane];
                    const int last = a.lane_off[e.lane] + a.lane_ncell[e.lane] - 1;
                    // the vehicle's speed is the cell's (conversion.py:55-60)
                    if (capflag[kc] & NS_CAP_OVERWRITTEN) {      // ... of the state before a later lane's deposit rewrote the cell
                        const float4 rc = caprec[kc];
                        float ar = 0.f, ay = 0.f;
                        glue_u_bwd(rc.x, rc.y, um, (float)gv, ar, ay);
                        Gr[last] += ar; Gy[last] += ay;
                    } else Gu[last] += (float)gv;
                    g_cap[kc] = ga;                     // ancillary a = the capacitor's level (value one vehicle length, gradient 1); the level
                                                        // the step leaves is a constant
                } else if (e.kind == NS_EV_CHANGE) {
                    const int m2 = a.lane_mslot[e.tgt], ms = a.lane_mslot[e.lane];
                    const size_t b2 = (size_t)m2 * cap;
                    const double gp = gPn[b2], gv = gVn[b2], ga = gA[b2];
                    ns_shift_out(gPn, gVn, gA, b2, n_bwd[m2]);
                    n_bwd[m2] -= 1;
                    const size_t hs = (size_t)ms * cap + n_bwd[ms];
                    gPn[hs] = gp; gVn[hs] = gv; gA[hs] = ga;
                    n_bwd[ms] += 1;
                } else if (e.kind == NS_EV_DESPAWN || e.kind == NS_EV_DEPOSIT) {
                    const int ms = a.lane_mslot[e.lane];
                    const size_t hs = (size_t)ms * cap + n_bwd[ms];
                    gPn[hs] = 0.; gVn[hs] = 0.; gA[hs] = 0.;
                    n_bwd[ms] += 1;
                } else if (e.kind == NS_EV_DEPCELL) {     // (behind its NS_EV_DEPOSIT in this order: the vehicle's slot is back)
                    const int ms = a.lane_mslot[e.lane];
                    const size_t hs = (size_t)ms * cap + n_bwd[ms] - 1;
                    const int c = e.cell;
                    float g_nr = Gr[c], g_speed = Gu[c];
                    glue_y_bwd(e.n_r, e.speed, um, Gy[c], g_nr, g_speed);
                    gA[hs] += (double)(g_nr * ((e.overlap / e.dx) / vlen));
                    gPn[hs] += (double)(g_nr * ((e.a / vlen) / e.dx) * e.f1);
                    gVn[hs] += (double)g_speed;
                    Gr[c] = g_nr; Gy[c] = 0.f; Gu[c] = 0.f;
                } else if (e.kind == NS_EV_CAPSERIAL) {    // a capacitor that read a cell a vehicle was deposited into in this very step
                    const int kc = a.lane_cslot[e.lane];
                    const int last = a.lane_off[e.lane] + a.lane_ncell[e.lane] - 1;
                    const float4 rc = caprec[kc];
                    const double gm = g_cap[kc] * (double)dtf;
                    Gr[last] += (float)(gm * (double)rc.z);
                    Gu[last] += (float)(gm * (double)rc.x);
                }
            }
        }
        __syncthreads();
        // ---- the capacitors' charges: level += (r u) dt (the level's cotangent passes on unchanged) ----
        for (int k = tid; k < a.ncap; k += B) {
            const int fl = capflag[k];
            if (!(fl & NS_CAP_CHARGED) || (fl & NS_CAP_SERIAL)) continue;
            const int l = a.cap_lanes[k];
            const int last = a.lane_off[l] + a.lane_ncell[l] - 1;
            const float4 rc = caprec[k];
            const double gm = g_cap[k] * (double)dtf;
            const float g_r = (float)(gm * (double)rc.z), g_u = (float)(gm * (double)rc.x);
            if (fl & NS_CAP_OVERWRITTEN) {          // a later lane's deposit rewrote the cell: the speed read was u(r, y) of the state before it
                float ar = g_r, ay = 0.f;
                glue_u_bwd(rc.x, rc.y, um, g_u, ar, ay);
                Gr[last] += ar; Gy[last] += ay;
            } else { Gr[last] += g_r; Gu[last] += g_u; }
        }
        __syncthreads();
    }
    // ---- (c) cells whose speed is u(r, y): fold the speed cotangent into (r, y) (a deposited cell's went to its vehicle above) ----
    for (int c = tid; c < C; c += B) {
        const float gu = Gu[c];
        if (gu != 0.f) { float gr = Gr[c], gy = Gy[c]; glue_u_bwd(Rn[c], Yn[c], um, gu, gr, gy); Gr[c] = gr; Gy[c] = gy; Gu[c] = 0.f; }
    }
    if (Lm == 0) return;
    __syncthreads();

    // ---- (d) IDM steps: g[i] = dEgo[i]^T g'[i] + dLeading[i-1]^T g'[i-1]; the head's virtual leader returns to the head and to the gap ----
    double *ghd = lds_d;                               // [Lm][2]
    const float4 *tape = ns_ptr<float4>(a, a.lo.idm_tape) + (size_t)t * plane;
    const int *nv_idm = ns_ptr<int>(a, a.lo.nv_idm) + (size_t)t * Lm;
    for (size_t idx = tid; idx < plane; idx += B) {
        const int m = (int)(idx / cap), i = (int)(idx - (size_t)m * cap);
        const int n = nv_idm[m];
        double gp = 0., gv = 0.;
        if (i < n) {
            const float4 q = tape[idx];                // (E2, E3, Ld2, Ld3); E0 = 1, E1 = dt, Ld0 = Ld1 = 0
            const double hp = gPn[idx], hv = gVn[idx];
            if (i == n - 1) {
                gp = hp * 1.0 + hv * (double)(q.x + q.z);
                gv = hp * (double)dtf + hv * (double)(q.y + q.w);
                ghd[2 * m] = hv * (double)q.z;
                ghd[2 * m + 1] = hv * (double)(-q.w);
            } else {
                gp = hp * 1.0 + hv * (double)q.x;
                gv = hp * (double)dtf + hv * (double)q.y;
            }
            if (i > 0) { const float4 f = tape[idx - 1]; const double fv = gVn[idx - 1]; gp += fv * (double)f.z; gv += fv * (double)f.w; }
        }
        gPp[idx] = gp; gVp[idx] = gv;
        if (i == 0 && n == 0) { ghd[2 * m] = 0.; ghd[2 * m + 1] = 0.; }
    }
    __syncthreads();
    if (tid == 0) for (int m = 0; m < Lm; m++) n_bwd[m] = nv_idm[m];      // (equal by construction; keeps a fault from spreading)

    // ---- (e) head gaps: Jacobian^T of (head_dp, head_dv) to the head vehicle, to its leader (another lane's tail) and to the signals ----
    const NsHeadGap *hgrow = ns_ptr<NsHeadGap>(a, a.lo.hg) + (size_t)t * Lm;
    double *g_act = ns_ptr<double>(a, a.lo.g_act);
    int *li_key = reinterpret_cast<int *>(lds_d + 2 * (size_t)Lm);          // compact lists: leader items [<= B], signal items [<= 3 B]
    double *li_p = lds_d + 2 * (size_t)Lm + (size_t)B * 2;                  // (keys take 4 B ints = 2 B doubles of room)
    double *li_v = li_p + B;
    double *si_v = li_v + B;
    int *si_key = li_key + B;
    for (int base = 0; base < Lm; base += B) {
        const int m = base + tid;
        NsHeadGap h; h.valid = 0; h.leader = -1; h.sig[0] = h.sig[1] = h.sig[2] = -1;
        double ghp = 0., ghv = 0.;
        if (m < Lm) {
            h = hgrow[m];
            if (h.valid) {
                ghp = ghd[2 * m]; ghv = ghd[2 * m + 1];
                const size_t hs = (size_t)m * cap + nv_idm[m] - 1;
                gPp[hs] += (double)h.jp[0] * ghp + (double)h.jv[0] * ghv;
                gVp[hs] += (double)h.jp[1] * ghp + (double)h.jv[1] * ghv;
            }
        }
        __syncthreads();
        // leaders: items compacted in lane order, every target lane adds the ones that name it
        const int has_l = (h.valid && h.leader >= 0) ? 1 : 0;
        int n_sig = 0;
        if (h.valid) for (int w = 0; w < 3; w++) n_sig += h.sig[w] >= 0;
        int tot_l, tot_s;
        const int pos_l = ns_block_scan<int>(has_l, scan_i, tot_l) - has_l;
        int pos_s = ns_block_scan<int>(n_sig, scan_i, tot_s) - n_sig;
        if (has_l) {
            li_key[pos_l] = h.leader;
            li_p[pos_l] = (double)h.jp[2] * ghp + (double)h.jv[2] * ghv;
            li_v[pos_l] = (double)h.jp[3] * ghp + (double)h.jv[3] * ghv;
        }
        if (h.valid) for (int w = 0; w < 3; w++) if (h.sig[w] >= 0) {
            float ds; int ai;
            (void)ns_lane_signal(a, action, t, h.sig[w], false, &ds, &ai);
            si_key[pos_s] = ai;
            si_v[pos_s] = ((double)h.jp[4 + w] * ghp + (double)h.jv[4 + w] * ghv) * (double)ds;
            pos_s++;
        }
        __syncthreads();
        if (tot_l) for (int m2 = tid; m2 < Lm; m2 += B) {
            double ap = 0., av = 0.; bool any = false;
            for (int k = 0; k < tot_l; k++) if (li_key[k] == m2) { ap += li_p[k]; av += li_v[k]; any = true; }
            if (any) { gPp[(size_t)m2 * cap] += ap; gVp[(size_t)m2 * cap] += av; }
        }
        if (tot_s) for (int q = tid; q < a.n_action; q += B) {
            double av = 0.; bool any = false;
            for (int k = 0; k < tot_s; k++) if (si_key[k] == q) { av += si_v[k]; any = true; }
            if (any) g_act[q] += av;
        }
        __syncthreads();
    }

    // ---- (f) admission undone: the admitted vehicle (position 0, speed 0: constants) leaves through the tail ----
    const int *adm = ns_ptr<int>(a, a.lo.adm) + (size_t)t * Lm;
    for (int m = tid; m < Lm; m += B) {
        if (adm[m]) { ns_shift_out(gPp, gVp, gA, (size_t)m * cap, n_bwd[m]); n_bwd[m] -= 1; }
    }
}
__global__ void __launch_bounds__(kNsBlock) ns_micro_bwd_kernel(NsArgs a, int t, const float *__restrict__ action, const float *__restrict__ g_reward) {
    extern __shared__ double lds_dyn[];
    ns_micro_bwd(a, t, action, g_reward, lds_dyn);
}

// =====================================================================================================================
// reverse, part 2: the lanes' ghost cotangents (dhts_macro_step_bwd's g_ghost) to the neighbours' edge cells, the stored ghosts and
// the action.  slot [L][2][4] = (cotangent for the source cell's r, for its u, the action partial, the action index as a float)
// =====================================================================================================================
__device__ __forceinline__ void ns_ghost_bwd_item(const NsArgs &a, int t, const float *__restrict__ action, int j) {
    const int L = a.L, C = a.C;
    if (j >= 2 * L) return;
    const int lane = j >> 1, side = j & 1;
    float *sl = ns_slots(a) + (size_t)j * 4;
    float add_r = 0.f, add_u = 0.f, a_val = 0.f; int a_key = -1;
    if (a.lane_macro[lane]) {
        const float um = a.um;
        const float *cur = ns_state(a, t);
        const NsRow rw = ns_row(a, t);
        const double *gg = ns_gghost(a) + ((size_t)a.lane_gpos[lane] * 2 + side) * 2;
        const float gg_r = (float)gg[0], gg_y = (float)gg[1];
        if (side == 0) {
            const int ls = rw.left_src[lane], lg = rw.left_gate[lane];
            if (ls >= 0) {
                const int last = a.lane_off[ls] + a.lane_ncell[ls] - 1;
                const float grn_r = cur[last], grn_u = cur[2 * C + last];
                float s = 1.f, ds = 0.f; int ai = -1;
                if (lg == -1) s = 0.f;
                else if (lg >= 0) s = ns_lane_signal(a, action, t, lg, false, &ds, &ai);
                const float fr = grn_r * s + 0.f * (1.0f - s), fu = grn_u * s + um * (1.0f - s);
                float g_fr = gg_r, g_fu = 0.f;
                glue_y_bwd(fr, fu, um, gg_y, g_fr, g_fu);
                add_r = g_fr * s; add_u = g_fu * s;
                if (ai >= 0) { a_val = (g_fr * grn_r + g_fu * (grn_u - um)) * ds; a_key = ai; }
            }
        } else {
            const int rs = rw.right_src[lane];
            const float *own_in = ns_ptr<float>(a, a.lo.own_hist) + (size_t)t * 2 * L;
            float *g_own = ns_ptr<float>(a, a.lo.g_own);
            const int first = rs < 0 ? 0 : a.lane_off[rs];
            const float grn_r = rs < 0 ? own_in[2 * lane] : cur[first];
            const float grn_u = rs < 0 ? own_in[2 * lane + 1] : cur[2 * C + first];
            float ds = 0.f; int ai = -1;
            const float sg = ns_lane_signal(a, action, t, lane, false, &ds, &ai);
            const float s2 = soft_switch(sg - 0.5f, kSigK);
            const float fr = s2 * grn_r + (1.0f - s2) * 1.0f, fu = s2 * grn_u + (1.0f - s2) * 0.0f;
            float g_fr = gg_r + g_own[2 * lane], g_fu = g_own[2 * lane + 1];       // the blended ghost is also the stored one
            glue_y_bwd(fr, fu, um, gg_y, g_fr, g_fu);
            if (rs >= 0) { add_r = g_fr * s2; add_u = g_fu * s2; g_own[2 * lane] = 0.f; g_own[2 * lane + 1] = 0.f; }
            else { g_own[2 * lane] = g_fr * s2; g_own[2 * lane + 1] = g_fu * s2; }
            if (ai >= 0) {
                const float g_s2 = g_fr * (grn_r - 1.0f) + g_fu * grn_u;
                a_val = g_s2 * soft_switch_grad(sg - 0.5f, kSigK) * ds; a_key = ai;
            }
        }
    }
    sl[0] = add_r; sl[1] = add_u; sl[2] = a_val; sl[3] = (float)a_key;
}
__global__ void ns_ghosts_bwd_kernel(NsArgs a, int t, const float *__restrict__ action) {
    ns_ghost_bwd_item(a, t, action, blockIdx.x * blockDim.x + threadIdx.x);
}

// thread = lane m: its edge cells take what the ghosts that looked at them left, in a fixed order (downstream lanes ascending for
// the last cell, upstream lanes ascending for the first); threads q < sq sum their intersection's action partials in slot order
__device__ __forceinline__ void ns_ghost_gather_item(const NsArgs &a, int t, int m) {
    const int L = a.L, C = a.C;
    const float *slot = ns_slots(a);
    float *G = ns_G(a, t);
    const NsRow rw = ns_row(a, t);
    if (m < L && a.lane_macro[m]) {
        const int first = a.lane_off[m], last = first + a.lane_ncell[m] - 1;
        float vr = 0.f, vu = 0.f;
        for (int e = a.nxt_ptr[m]; e < a.nxt_ptr[m + 1]; ++e) {
            const int b = a.nxt_idx[e];
            if (a.lane_macro[b] && rw.left_src[b] == m) { vr += slot[(size_t)(2 * b) * 4]; vu += slot[(size_t)(2 * b) * 4 + 1]; }
        }
        float wr = 0.f, wu = 0.f;
        for (int e = a.prv_ptr[m]; e < a.prv_ptr[m + 1]; ++e) {
            const int b = a.prv_idx[e];
            if (a.lane_macro[b] && rw.right_src[b] == m) { wr += slot[(size_t)(2 * b + 1) * 4]; wu += slot[(size_t)(2 * b + 1) * 4 + 1]; }
        }
        if (first == last) { G[first] += vr + wr; G[2 * C + first] += vu + wu; }
        else { G[last] += vr; G[2 * C + last] += vu; G[first] += wr; G[2 * C + first] += wu; }
    }
    if (m < a.sq) {
        double v = 0.; int key = -1;
        for (int k = a.inter_ptr[m]; k < a.inter_ptr[m + 1]; ++k) {
            const float *sl = slot + (size_t)a.inter_idx[k] * 4;
            if (sl[3] >= 0.f) { v += (double)sl[2]; key = (int)sl[3]; }
        }
        if (key >= 0) ns_ptr<double>(a, a.lo.g_act)[key] += v;
    }
}
__global__ void ns_ghosts_gather_kernel(NsArgs a, int t) {
    ns_ghost_gather_item(a, t, blockIdx.x * blockDim.x + threadIdx.x);
}

// =====================================================================================================================
// PERSISTENT form (round 5): the same step -- the very device functions above -- inside ONE kernel per direction, one workgroup per
// network replica, all T steps, workgroup barriers where the stepwise form has kernel boundaries.  The workgroup's threads loop
// over the network's items (ghosts, interfaces, cells: a few per thread for the grids the reference builds); the state of a step
// is the history row in HBM (L2-resident: tens of KB), interface results and the blocks dqs[c][3][2][2] (dMacroLane._backward,
// dmacro_lane.py:96-132) go through the replica's workspace.  No cross-workgroup exchange exists, so nothing spins; replicas are
// independent workgroups (BASELINE config 5's pattern for networks beyond the fused kernels' one-item-per-thread limits).
// A step costs its phases' latencies (~10 us forward + reverse for 360 lanes / 2 124 cells) instead of ~100 us of launches.
// =====================================================================================================================
__device__ __forceinline__ NsArgs ns_replica_args(const NsArgs &a0, int rep) {
    NsArgs a = a0;
    a.hist += (size_t)rep * (a.T + 1) * 4 * (a.C > 0 ? a.C : 1);
    a.queue += (size_t)rep * a.T * a.L;
    if (a.reward) a.reward += (size_t)rep * 2;
    if (a.counts) a.counts += (size_t)rep * 4;
    a.ws += (size_t)rep * a.lo.total;
    const size_t ts = (size_t)rep * (size_t)a.table_stride;
    a.left_src += ts; a.left_gate += ts; a.right_src += ts; a.conv_next += ts; a.schedule += ts;
    if (a.draws) a.draws += (size_t)rep * (size_t)a.draws_stride;
    return a;
}

// One cell of step t: BOTH its interfaces (ARZ.riemann_solve + the two Jacobian products each, redone by the neighbouring cell's
// thread: a solve costs less than handing its result over), Godunov update, float32 glue, the cell's three blocks (MacroLane.forward,
// _macro_lane.py:83-146; dMacroLane._backward, dmacro_lane.py:96-132) -- the operations of the straight-lane operator's kernel
__device__ __forceinline__ void ns_cell_item(const NsArgs &a, int t, int c, int &fault_step, int &fault_lane, int &fault_index) {
    const int C = a.C;
    const int l = a.cell_lane[c];
    const int off = a.lane_off[l], n = a.lane_ncell[l];
    const int k = c - off;
    const double dx = a.lane_dx[l];
    const double cc = a.dt_d / dx;
    const float *cur = ns_state(a, t);
    float *nxt = ns_state(a, t + 1);
    const float *gh = ns_ghosts(a) + (size_t)a.lane_gpos[l] * 8;
    const double r0 = cur[c], y0 = cur[C + c], u0 = cur[2 * C + c], q0 = cur[3 * C + c];
    double rL, yL, uL, qL, rR, yR, uR, qR;
    if (k == 0) { rL = gh[0]; yL = gh[1]; uL = gh[2]; qL = gh[3]; }
    else { rL = cur[c - 1]; yL = cur[C + c - 1]; uL = cur[2 * C + c - 1]; qL = cur[3 * C + c - 1]; }
    if (k == n - 1) { rR = gh[4]; yR = gh[5]; uR = gh[6]; qR = gh[7]; }
    else { rR = cur[c + 1]; yR = cur[C + c + 1]; uR = cur[2 * C + c + 1]; qR = cur[3 * C + c + 1]; }
    IfaceConst kc;
    kc.set_um(a.um_d); kc.set_grid(a.dt_d, dx);
    Iface fl, fr;
    arz_interface(rL, yL, uL, qL, r0, y0, u0, q0, kc, fl);
    arz_interface(r0, y0, u0, q0, rR, yR, uR, qR, kc, fr);
    if (fault_step < 0 && (fl.cfl_bad || (k == n - 1 && fr.cfl_bad))) { fault_step = t; fault_lane = l; fault_index = fl.cfl_bad ? k : n; }
    const float nr = (float)(r0 + (fl.Fr - fr.Fr) * cc);
    const float ny = (float)(y0 + (fl.Fy - fr.Fy) * cc);
    float nu, nq;
    glue_from_r_y(nr, ny, a.um, nu, nq);
    nxt[c] = nr; nxt[C + c] = ny; nxt[2 * C + c] = nu; nxt[3 * C + c] = nq;
    if (!a.hard) {
        const float cf = (float)cc, ncf = (float)(-cc);
        float4 d0, d1, d2;
        d0.x = ncf * (-fl.A[0]); d0.y = ncf * (-fl.A[1]); d0.z = ncf * (-fl.A[2]); d0.w = ncf * (-fl.A[3]);
        d2.x = ncf * fr.B[0]; d2.y = ncf * fr.B[1]; d2.z = ncf * fr.B[2]; d2.w = ncf * fr.B[3];
        d1.x = 1.f - cf * (fr.A[0] - fl.B[0]); d1.y = 0.f - cf * (fr.A[1] - fl.B[1]);
        d1.z = 0.f - cf * (fr.A[2] - fl.B[2]); d1.w = 1.f - cf * (fr.A[3] - fl.B[3]);
        float4 *tp = ns_ptr<float4>(a, a.lo.ptape) + ((size_t)t * C + c) * 3;
        tp[0] = d0; tp[1] = d1; tp[2] = d2;
    }
}
// reverse of the cells of step t: g[c] <- dqs[c][1]^T g[c] + dqs[c-1][2]^T g[c-1] + dqs[c+1][0]^T g[c+1] inside a lane
// (dMacroForwardLayer.backward, dmacro_lane.py:277-309); the lane's edge cells leave the ghosts' cotangents
__device__ __forceinline__ void ns_cell_bwd_item(const NsArgs &a, int t, int c) {
    const int C = a.C;
    const int l = a.cell_lane[c];
    const int off = a.lane_off[l], n = a.lane_ncell[l];
    const int k = c - off;
    const float *Gn = ns_G(a, t + 1);
    float *Gp = ns_G(a, t);
    const float4 *tp = ns_ptr<float4>(a, a.lo.ptape) + ((size_t)t * C + c) * 3;
    const float gr = Gn[c], gy = Gn[C + c];
    const float4 d0 = tp[0], d1 = tp[1], d2 = tp[2];
    float4 e2 = make_float4(0.f, 0.f, 0.f, 0.f), e0 = e2;
    if (k > 0) e2 = tp[-1];              // dqs[c - 1][2]
    if (k < n - 1) e0 = tp[3];           // dqs[c + 1][0]
    float vr = dot2(d1.x, gr, d1.z, gy), vy = dot2(d1.y, gr, d1.w, gy);
    float c2r = 0.f, c2y = 0.f, c0r = 0.f, c0y = 0.f;
    if (k > 0) { const float hr = Gn[c - 1], hy = Gn[C + c - 1]; c2r = dot2(e2.x, hr, e2.z, hy); c2y = dot2(e2.y, hr, e2.w, hy); }
    if (k < n - 1) { const float hr = Gn[c + 1], hy = Gn[C + c + 1]; c0r = dot2(e0.x, hr, e0.z, hy); c0y = dot2(e0.y, hr, e0.w, hy); }
    const float pr = (vr + c2r) + c0r, py = (vy + c2y) + c0y;
    Gp[c] = pr; Gp[C + c] = py; Gp[2 * C + c] = 0.f;
    double *gg = ns_gghost(a) + (size_t)a.lane_gpos[l] * 4;
    if (k == 0) { gg[0] = (double)dot2(d0.x, gr, d0.z, gy); gg[1] = (double)dot2(d0.y, gr, d0.w, gy); }
    if (k == n - 1) { gg[2] = (double)dot2(d2.x, gr, d2.z, gy); gg[3] = (double)dot2(d2.y, gr, d2.w, gy); }
    if (!(isfinite(pr) && isfinite(py))) net_fault(a.err, DHTS_FAULT_NAN, t, l, k);
}

// The static tables of a network (per lane, per interface, per cell, adjacency) staged in LDS for a persistent kernel: every item of
// every phase starts with a chain of dependent look-ups (item -> lane -> offsets -> state), each link a global-memory latency
// otherwise.  The functions above read them through NsArgs' pointers, which are simply redirected.
__host__ __device__ inline size_t ns_al16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t ns_stage_bytes(int L, int C, int NI, int Lm, int ncap, int n_edges, int sq, int n_islots, bool sources) {
    return 8 * ns_al16(4 * (size_t)L) + (sources ? ns_al16(4 * (size_t)L) : 0) + 2 * ns_al16(8 * (size_t)L) + ns_al16(4 * (size_t)NI) +
           ns_al16(4 * (size_t)(C > 0 ? C : 1)) + ns_al16(4 * (size_t)(Lm > 0 ? Lm : 1)) + ns_al16(4 * (size_t)(ncap > 0 ? ncap : 1)) +
           2 * ns_al16(4 * (size_t)(L + 1)) + 2 * ns_al16(4 * (size_t)(n_edges > 0 ? n_edges : 1)) + ns_al16(4 * (size_t)(sq + 1)) +
           ns_al16(4 * (size_t)(n_islots > 0 ? n_islots : 1));
}
template <typename T>
__device__ __forceinline__ const T *ns_stage(const T *src, size_t n, char *&p) {
    T *dst = reinterpret_cast<T *>(p);
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    p += ns_al16(n * sizeof(T));
    return dst;
}
__device__ __forceinline__ void ns_stage_tables(NsArgs &a, char *p) {
    if (!a.stage) return;
    const size_t L = a.L;
    a.lane_ncell = ns_stage(a.lane_ncell, L, p); a.lane_off = ns_stage(a.lane_off, L, p); a.sig_kind = ns_stage(a.sig_kind, L, p);
    a.inter = ns_stage(a.inter, L, p); a.lane_macro = ns_stage(a.lane_macro, L, p); a.lane_gpos = ns_stage(a.lane_gpos, L, p);
    a.lane_mslot = ns_stage(a.lane_mslot, L, p); a.lane_cslot = ns_stage(a.lane_cslot, L, p);
    if (a.lane_source) a.lane_source = ns_stage(a.lane_source, L, p);
    a.lane_dx = ns_stage(a.lane_dx, L, p); a.lane_len = ns_stage(a.lane_len, L, p);
    a.if_lane = ns_stage(a.if_lane, (size_t)a.NI, p); a.cell_lane = ns_stage(a.cell_lane, (size_t)(a.C > 0 ? a.C : 1), p);
    a.micro_lanes = ns_stage(a.micro_lanes, (size_t)(a.Lm > 0 ? a.Lm : 1), p); a.cap_lanes = ns_stage(a.cap_lanes, (size_t)(a.ncap > 0 ? a.ncap : 1), p);
    a.nxt_ptr = ns_stage(a.nxt_ptr, L + 1, p); a.prv_ptr = ns_stage(a.prv_ptr, L + 1, p);
    a.nxt_idx = ns_stage(a.nxt_idx, (size_t)(a.n_edges > 0 ? a.n_edges : 1), p); a.prv_idx = ns_stage(a.prv_idx, (size_t)(a.n_edges > 0 ? a.n_edges : 1), p);
    a.inter_ptr = ns_stage(a.inter_ptr, (size_t)a.sq + 1, p); a.inter_idx = ns_stage(a.inter_idx, (size_t)(a.n_islots > 0 ? a.n_islots : 1), p);
}

// what a persistent kernel carves out of its dynamic LDS behind the phases' scratch (host and device agree through this plan)
struct NsPlan { int scratch, tables, st, gl, rows, misc; };       // byte sizes (0 = not staged)
__device__ __forceinline__ void ns_carve(NsArgs &a, char *lds, const NsPlan &pl) {
    char *p = lds + pl.scratch;
    if (pl.tables) { a.stage = 1; char *q = p; ns_stage_tables(a, q); p += pl.tables; } else a.stage = 0;
    a.st = nullptr; a.gl = nullptr; a.row_i = nullptr; a.row_d = nullptr; a.gh = nullptr; a.sl = nullptr; a.gg = nullptr;
    if (pl.st) { a.st = reinterpret_cast<float *>(p); p += pl.st; }
    if (pl.gl) { a.gl = reinterpret_cast<float *>(p); p += pl.gl; }
    if (pl.rows) {
        a.row_d = reinterpret_cast<double *>(p); a.row_i = reinterpret_cast<int32_t *>(p + ns_al16(16 * (size_t)a.L));
        p += pl.rows;
    }
    if (pl.misc) {
        const size_t lg = ns_al16(32 * (size_t)a.L);
        a.gg = reinterpret_cast<double *>(p); a.gh = reinterpret_cast<float *>(p + lg); a.sl = reinterpret_cast<float *>(p + 2 * lg);
    }
}
// the table rows of step r into their LDS copy (r & 1): every thread fetches its elements (registers `v`), stores them later
struct NsRowRegs { int32_t i[4]; double d; };
__device__ __forceinline__ void ns_rows_fetch(const NsArgs &a, const NsArgs &g, int r, int l, NsRowRegs &v) {      // g: the global pointers
    const size_t o = (size_t)r * a.L + l;
    v.i[0] = g.left_src[o]; v.i[1] = g.left_gate[o]; v.i[2] = g.right_src[o]; v.i[3] = g.conv_next[o]; v.d = g.schedule[o];
}
__device__ __forceinline__ void ns_rows_store(const NsArgs &a, int r, int l, const NsRowRegs &v) {
    int32_t *b = a.row_i + (size_t)(r & 1) * 4 * a.L;
    b[l] = v.i[0]; b[a.L + l] = v.i[1]; b[2 * a.L + l] = v.i[2]; b[3 * a.L + l] = v.i[3];
    a.row_d[(size_t)(r & 1) * a.L + l] = v.d;
}

__global__ void __launch_bounds__(kNsBlock) ns_persist_fwd_kernel(NsArgs a0, const float *__restrict__ action_all, NsPlan pl) {
    extern __shared__ double lds_p[];
    // the replica's arguments -- ~130 pointers and offsets, some redirected into LDS -- live in LDS themselves: as a local copy they
    // cost 487 scalar-register spills and 79 vector ones (scratch traffic inside every phase)
    __shared__ NsArgs ag_s, a_s;
    if (threadIdx.x == 0) { ag_s = ns_replica_args(a0, blockIdx.x); a_s = ag_s; }
    __syncthreads();
    {
        NsArgs tmp = a_s;
        ns_carve(tmp, reinterpret_cast<char *>(lds_p), pl);       // (every thread copies its share of the tables)
        __syncthreads();
        if (threadIdx.x == 0) a_s = tmp;
        __syncthreads();
    }
    const NsArgs &ag = ag_s;                                    // the global pointers
    const NsArgs &a = a_s;
    const float *action = action_all + (size_t)blockIdx.x * a0.n_action;
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a0.L, C = a0.C, T = a0.T;
    // the episode's running state: empty road (MacroLane.__init__), stored ghosts (0, u_max), no vehicles, counters zero
    {
        float *s0 = ns_state(a, 0);
        for (int i = tid; i < C; i += B) {
            s0[i] = 0.f; s0[C + i] = 0.f; s0[2 * C + i] = a.um; s0[3 * C + i] = a.um;
            if (a.st) { a.hist[i] = 0.f; a.hist[C + i] = 0.f; a.hist[2 * C + i] = a.um; a.hist[3 * C + i] = a.um; }
        }
    }
    for (int i = tid; i < L; i += B) { float *o = ns_ptr<float>(a, a.lo.own_hist); o[2 * i] = 0.f; o[2 * i + 1] = a.um; }
    {
        unsigned *z = ns_ptr<unsigned>(a, a.lo.P);
        const size_t nz = (a.lo.counters + sizeof(NsCounters) - a.lo.P) / 4;
        for (size_t i = tid; i < nz; i += B) z[i] = 0u;
    }
    if (a.row_i) for (int l = tid; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, ag, 0, l, v); ns_rows_store(a, 0, l, v); }
    __syncthreads();
    int fault_step = -1, fault_lane = 0, fault_index = 0;
#ifdef DHTS_NS_STAMPS
    long long st_[6] = {0, 0, 0, 0, 0, 0}, st_last_ = __builtin_amdgcn_s_memtime();
#define NS_STAMP(i) { const long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_last_; st_last_ = n_; }
#else
#define NS_STAMP(i)
#endif
    for (int t = 0; t < T; ++t) {
        // the next step's table rows: fetched now (one lane per thread; further lanes of wider networks at the end of the step)
        NsRowRegs nx;
        const bool pre = a.row_i && t + 1 < T && tid < L;
        if (pre) ns_rows_fetch(a, ag, t + 1, tid, nx);
        for (int j = tid; j < 2 * L; j += B) ns_ghost_fwd_item(a, t, action, j);
        NS_STAMP(0)
        ns_micro_fwd(a, t, action, reinterpret_cast<float *>(lds_p));
        __syncthreads();
        NS_STAMP(1)
        for (int c = tid; c < C; c += B) ns_cell_item(a, t, c, fault_step, fault_lane, fault_index);
        __syncthreads();
        NS_STAMP(2)
        ns_convert(a, t, reinterpret_cast<int *>(lds_p));
        NS_STAMP(3)
        if (a.st) {                     // the committed state to the history (what the reverse sweep and the callers read)
            const float *sn = ns_state(a, t + 1);
            float *hn = a.hist + (size_t)(t + 1) * 4 * C;
            for (int i = tid; i < 4 * C; i += B) hn[i] = sn[i];
        }
        if (pre) ns_rows_store(a, t + 1, tid, nx);
        if (a.row_i && t + 1 < T) for (int l = tid + B; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, ag, t + 1, l, v); ns_rows_store(a, t + 1, l, v); }
        __syncthreads();
        NS_STAMP(4)
    }
#ifdef DHTS_NS_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        printf("ns_persist_fwd cycles per step: ghosts %lld | micro boundary + IDM %lld | cells %lld | hand-offs + loss %lld | flush %lld\n",
               st_[0] / T, st_[1] / T, st_[2] / T, st_[3] / T, st_[4] / T);
#endif
    if (fault_step >= 0) net_fault(a.err, DHTS_FAULT_CFL, fault_step, fault_lane, fault_index);
    ns_reward(a, reinterpret_cast<float *>(lds_p));
}

__global__ void __launch_bounds__(kNsBlock) ns_persist_bwd_kernel(NsArgs a0, const float *__restrict__ action_all, const float *__restrict__ g_reward,
                                                                float *__restrict__ g_action_all, NsPlan pl) {
    extern __shared__ double lds_p[];
    __shared__ NsArgs ag_s, a_s;                                // (see ns_persist_fwd_kernel)
    if (threadIdx.x == 0) { ag_s = ns_replica_args(a0, blockIdx.x); a_s = ag_s; }
    __syncthreads();
    {
        NsArgs tmp = a_s;
        ns_carve(tmp, reinterpret_cast<char *>(lds_p), pl);
        __syncthreads();
        if (threadIdx.x == 0) a_s = tmp;
        __syncthreads();
    }
    const NsArgs &ag = ag_s;
    const NsArgs &a = a_s;
    const float *action = action_all + (size_t)blockIdx.x * a0.n_action;
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a0.L, C = a0.C, T = a0.T, Lm = a0.Lm;
    {   // cotangents start at zero; the lanes hold what the forward left
        unsigned *z = ns_ptr<unsigned>(a, a.lo.G);
        const size_t nz = (a.lo.n_bwd - a.lo.G) / 4;
        for (size_t i = tid; i < nz; i += B) z[i] = 0u;
        if (a.gl) for (int i = tid; i < 6 * C; i += B) a.gl[i] = 0.f;
        int *n_bwd = ns_ptr<int>(a, a.lo.n_bwd);
        const int *lane_n = ns_ptr<int>(a, a.lo.lane_n);
        for (int m = tid; m < Lm; m += B) n_bwd[m] = lane_n[m];
    }
    if (a.st) {                          // rows T and T - 1 of the history
        for (int r = T; r >= T - 1 && r >= 0; --r) {
            float *sr = ns_state(a, r);
            const float *hr = a.hist + (size_t)r * 4 * C;
            for (int i = tid; i < 4 * C; i += B) sr[i] = hr[i];
        }
    }
    if (a.row_i) for (int l = tid; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, ag, T - 1, l, v); ns_rows_store(a, T - 1, l, v); }
    __syncthreads();
    const int m_items = L > a.sq ? L : a.sq;
#ifdef DHTS_NS_STAMPS
    long long st_[6] = {0, 0, 0, 0, 0, 0}, st_last_ = __builtin_amdgcn_s_memtime();
#endif
    for (int t = T - 1; t >= 0; --t) {
        NsRowRegs nx;
        const bool pre = a.row_i && t >= 1 && tid < L;
        if (pre) ns_rows_fetch(a, ag, t - 1, tid, nx);
        ns_micro_bwd(a, t, action, g_reward ? g_reward + blockIdx.x : nullptr, lds_p);
        __syncthreads();
        NS_STAMP(0)
        for (int c = tid; c < C; c += B) ns_cell_bwd_item(a, t, c);
        __syncthreads();
        NS_STAMP(1)
        if (C > 0) {
            for (int j = tid; j < 2 * L; j += B) ns_ghost_bwd_item(a, t, action, j);
            __syncthreads();
            NS_STAMP(2)
            for (int m = tid; m < m_items; m += B) ns_ghost_gather_item(a, t, m);
            NS_STAMP(3)
        }
        // the state before step t - 1 (row t - 1) into the copy row t + 1 leaves; the rows of step t - 1
        if (a.st && t >= 1) {
            float *sr = ns_state(a, t - 1);
            const float *hr = a.hist + (size_t)(t - 1) * 4 * C;
            for (int i = tid; i < 4 * C; i += B) sr[i] = hr[i];
        }
        if (pre) ns_rows_store(a, t - 1, tid, nx);
        if (a.row_i && t >= 1) for (int l = tid + B; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, ag, t - 1, l, v); ns_rows_store(a, t - 1, l, v); }
        __syncthreads();
        NS_STAMP(4)
    }
#ifdef DHTS_NS_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        printf("ns_persist_bwd cycles per step: taps + events + fold + IDM + head gaps %lld | cells %lld | ghosts %lld | gather %lld | state %lld\n",
               st_[0] / T, st_[1] / T, st_[2] / T, st_[3] / T, st_[4] / T);
#endif
    float *g_action = g_action_all + (size_t)blockIdx.x * a.n_action;
    const double *g_act = ns_ptr<double>(a, a.lo.g_act);
    for (int q = tid; q < a.n_action; q += B) g_action[q] = (float)g_act[q];
}

__global__ void ns_finish_bwd_kernel(NsArgs a, float *__restrict__ g_action) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < a.n_action) g_action[q] = (float)ns_ptr<double>(a, a.lo.g_act)[q];
}

// initial state: empty road (MacroLane.__init__: r = y = 0, u = u_eq = u_max), stored ghosts (0, u_max), no vehicles
__global__ void ns_init_fwd_kernel(NsArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.C) { a.hist[i] = 0.f; a.hist[a.C + i] = 0.f; a.hist[2 * a.C + i] = a.um; a.hist[3 * a.C + i] = a.um; }
    if (i < a.L) { float *o = ns_ptr<float>(a, a.lo.own_hist); o[2 * i] = 0.f; o[2 * i + 1] = a.um; }
}

}  // namespace dhts

using namespace dhts;

static inline int ns_max_events(const dhts_net_desc *d, const dhts_netstep_tables *t) {
    return t->max_events > 0 ? t->max_events : 8 * d->n_steps + 64;
}

static NsLayout ns_layout(const dhts_net_desc *d, const dhts_netstep_tables *t) {
    NsLayout o;
    const size_t L = d->n_lanes, C = d->n_cells, T = d->n_steps, Lm = t->hyb.n_micro, cap = t->hyb.lane_capacity > 0 ? t->hyb.lane_capacity : 16;
    const size_t ncap = t->n_caps, plane = Lm * cap;
    size_t tape_step = 0, Lg = 0;
    for (int g = 0; g < t->n_groups; ++g) {
        dhts_macro_desc md = {t->groups[g].n_lanes, t->groups[g].n_cells, d->dt, t->groups[g].dx, d->u_max};
        tape_step += dhts_macro_step_tape_bytes(&md) / sizeof(float);
        Lg += t->groups[g].n_lanes;
    }
    o.tape_step = tape_step;
    size_t p = 0;
    auto R = [&](size_t bytes) { size_t r = p; p += ns_up(bytes); return r; };
    o.ghost = R(sizeof(float) * Lg * 8); o.g_ghost = R(sizeof(double) * Lg * 4); o.slot = R(sizeof(float) * L * 8);
    o.own_hist = R(sizeof(float) * (T + 1) * 2 * L);
    o.tape = R(t->persistent ? 0 : sizeof(float) * T * tape_step);
    o.P = R(sizeof(float) * 2 * plane); o.V = R(sizeof(float) * 2 * plane); o.A = R(sizeof(float) * plane);
    o.vroute = R(sizeof(int) * plane); o.vcur = R(sizeof(int) * plane); o.lane_n = R(sizeof(int) * Lm); o.rused = R(sizeof(int) * L);
    o.capv = R(sizeof(float) * ncap); o.counters = R(sizeof(NsCounters));
    o.sigS = R(sizeof(double) * (T * Lm + 1)); o.lossS = R(sizeof(double) * (T * (C + plane) + 1));
    o.hg = R(sizeof(NsHeadGap) * T * Lm); o.idm_tape = R(sizeof(float4) * T * plane);
    o.nv_idm = R(sizeof(int) * T * Lm); o.adm = R(sizeof(int) * T * Lm); o.nv_post = R(sizeof(int) * T * Lm);
    o.vx = R(sizeof(float) * T * plane); o.kc_cell = R(sizeof(float) * T * C); o.kc_veh = R(sizeof(float) * T * plane);
    o.caprec = R(sizeof(float4) * (T * ncap + 1)); o.capflag = R(sizeof(int) * (T * ncap + 1));
    o.ev = R(sizeof(NsEvent) * (size_t)ns_max_events(d, t)); o.ev_off = R(sizeof(int) * (T + 2));
    o.G = R(sizeof(float) * 2 * 3 * C);
    o.gP = R(sizeof(double) * 2 * plane); o.gV = R(sizeof(double) * 2 * plane); o.gA = R(sizeof(double) * plane);
    o.g_cap = R(sizeof(double) * ncap); o.g_own = R(sizeof(float) * 2 * L); o.g_act = R(sizeof(double) * (size_t)d->n_action);
    o.n_bwd = R(sizeof(int) * Lm); o.ghd = R(16);
    // the persistent kernels' own scratch and Jacobian tape (blocks dqs[c][3][2][2] per cell-step: 48 B, dmacro_lane.py:50-56)
    const size_t NI = C + Lg;
    o.pF = o.pAB = p; (void)NI;
    o.ptape = R(t->persistent ? sizeof(float) * 12 * T * C : 0);
    if (t->persistent) { o.tape = o.ptape; }          // (the per-group operator tape is not used then)
    o.total = p;
    return o;
}

static bool ns_ok(const dhts_net_desc *d, const dhts_netstep_tables *t) {
    if (!d || !t) return false;
    const dhts_hybrid_tables &h = t->hyb;
    const int cap = h.lane_capacity > 0 ? h.lane_capacity : 16;
    return (d->n_replicas == 1 || (t->persistent && d->n_replicas > 1)) && (!t->persistent || d->n_cells == 0 || (t->if_lane && t->cell_lane)) &&
           d->n_lanes > 0 && d->n_cells >= 0 && d->n_steps > 0 && d->n_inter_sq > 0 && d->frames_per_phase > 0 &&
           d->n_action >= d->n_inter_sq && d->dt > 0 && d->u_max > 0 && d->vehicle_length > 0 && cap <= 1024 && h.n_micro >= 0 &&
           h.net.lane_ncell && h.net.lane_off && h.net.sig_kind && h.net.inter && h.net.lane_dx && h.net.left_src && h.net.left_gate &&
           h.net.right_src && h.net.schedule && h.net.nxt_ptr && h.net.nxt_idx && h.net.prv_ptr && h.net.prv_idx && h.lane_macro &&
           h.lane_len && h.conv_next && t->lane_gpos && (t->n_groups == 0 || t->groups) && t->n_groups >= 0 &&
           (h.n_micro == 0 || (t->micro_lanes && t->lane_mslot && h.routes && h.route_ptr && h.route_stride > 0 && h.route_stride <= kNsRouteMax)) &&
           (t->n_caps == 0 || (t->cap_lanes && t->lane_cslot)) && t->inter_ptr && t->inter_idx && (d->n_cells == 0 || t->n_groups > 0) &&
           (h.lane_source == nullptr || (h.draws != nullptr && h.n_draws > 0));
}

static NsArgs ns_args(const dhts_net_desc *d, const dhts_netstep_tables *t, int hard, float *hist, float *queue, float *reward, int32_t *counts,
                      void *ws, dhts_error *err) {
    NsArgs a;
    const dhts_hybrid_tables &h = t->hyb;
    a.L = d->n_lanes; a.C = d->n_cells; a.T = d->n_steps; a.sq = d->n_inter_sq; a.F = d->frames_per_phase; a.n_action = d->n_action;
    a.Lm = h.n_micro; a.cap = h.lane_capacity > 0 ? h.lane_capacity : 16; a.ncap = t->n_caps; a.hard = hard;
    a.route_stride = h.route_stride; a.n_draws = h.n_draws; a.max_events = ns_max_events(d, t); a.loss_steps = h.loss_steps;
    a.um = (float)d->u_max; a.dtf = (float)d->dt; a.vlen = (float)d->vehicle_length; a.s0f = (float)d->static_speed;
    a.um_d = d->u_max; a.dt_d = d->dt; a.vlen_d = d->vehicle_length;
    a.lane_ncell = h.net.lane_ncell; a.lane_off = h.net.lane_off; a.sig_kind = h.net.sig_kind; a.inter = h.net.inter;
    a.lane_macro = h.lane_macro; a.lane_gpos = t->lane_gpos; a.micro_lanes = t->micro_lanes; a.lane_mslot = t->lane_mslot;
    a.cap_lanes = t->cap_lanes; a.lane_cslot = t->lane_cslot; a.lane_source = h.lane_source;
    a.lane_dx = h.net.lane_dx; a.lane_len = h.lane_len;
    a.left_src = h.net.left_src; a.left_gate = h.net.left_gate; a.right_src = h.net.right_src; a.conv_next = h.conv_next;
    a.schedule = h.net.schedule; a.routes = h.routes; a.route_ptr = h.route_ptr; a.draws = h.draws;
    a.nxt_ptr = h.net.nxt_ptr; a.nxt_idx = h.net.nxt_idx; a.prv_ptr = h.net.prv_ptr; a.prv_idx = h.net.prv_idx;
    a.inter_ptr = t->inter_ptr; a.inter_idx = t->inter_idx;
    a.if_lane = t->if_lane; a.cell_lane = t->cell_lane; a.table_stride = h.net.replica_stride; a.draws_stride = h.draws_stride;
    a.n_edges = h.net.n_edges; a.n_islots = t->n_inter_slots; a.stage = 0;
    a.st = nullptr; a.gl = nullptr; a.gh = nullptr; a.sl = nullptr; a.gg = nullptr; a.row_i = nullptr; a.row_d = nullptr;
    { int lg = 0; for (int g = 0; g < t->n_groups; ++g) lg += t->groups[g].n_lanes; a.NI = d->n_cells + lg; }
    a.ws = reinterpret_cast<char *>(ws); a.lo = ns_layout(d, t);
    a.hist = hist; a.queue = queue; a.reward = reward; a.counts = counts; a.err = err;
    return a;
}

// What fits into a workgroup's 160 KB beside the phases' scratch, in the order of what it buys: the per-step table rows and the
// ghost / slot arrays (small, read by every item), the static tables, the cotangent planes (reverse sweep) resp. the state rows,
// then the other of the two.
static NsPlan ns_plan(const NsArgs &a, size_t scratch, bool bwd) {
    NsPlan pl = {(int)scratch, 0, 0, 0, 0, 0};
    size_t left = 156 * 1024 - scratch;
    auto take = [&](size_t bytes, int &slot) { if (bytes && bytes <= left) { slot = (int)bytes; left -= bytes; } };
    take(ns_al16(16 * (size_t)a.L) + ns_al16(32 * (size_t)a.L), pl.rows);
    take(3 * ns_al16(32 * (size_t)a.L), pl.misc);
    take(ns_stage_bytes(a.L, a.C, a.NI, a.Lm, a.ncap, a.n_edges, a.sq, a.n_islots, a.lane_source != nullptr), pl.tables);
    const size_t st = ns_al16(sizeof(float) * 8 * (size_t)a.C), gl = ns_al16(sizeof(float) * 6 * (size_t)a.C);
    if (a.C > 0) {
        if (bwd) { take(gl, pl.gl); take(st, pl.st); }
        else take(st, pl.st);
    }
    return pl;
}

extern "C" {

size_t dhts_netstep_workspace_bytes(const dhts_net_desc *d, const dhts_netstep_tables *t) {
    if (!ns_ok(d, t)) return 0;
    return ns_layout(d, t).total * (size_t)d->n_replicas;
}

int dhts_netstep_rollout_fwd(const dhts_net_desc *d, const dhts_netstep_tables *t, int hard, const float *action, float *hist,
                             float *queue, float *reward, int32_t *counts, void *workspace, dhts_error *err, void *stream) {
    if (!ns_ok(d, t) || !action || !hist || !queue || !reward || !counts || !workspace) return DHTS_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    NsArgs a = ns_args(d, t, hard, hist, queue, reward, counts, workspace, err);
    const int L = a.L, C = a.C, T = a.T, Lm = a.Lm;
    const size_t plane = (size_t)Lm * a.cap;
    char *ws = a.ws;
    if (t->persistent) {
        size_t lds = sizeof(float) * 2 * (size_t)(Lm > 0 ? Lm : 1);              // head gaps | candidate flags | the reward's partial sums
        if (sizeof(float) * 2 * (size_t)L > lds) lds = sizeof(float) * 2 * (size_t)L;
        const NsPlan pl = ns_plan(a, ns_al16(lds), false);
        lds = (size_t)pl.scratch + pl.tables + pl.st + pl.gl + pl.rows + pl.misc;
        if (lds > 160 * 1024 ||
            (lds > 48 * 1024 && hipFuncSetAttribute((const void *)ns_persist_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess))
            return DHTS_E_INVALID;
        ns_persist_fwd_kernel<<<d->n_replicas, kNsBlock, lds, st>>>(a, action, pl);
        return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
    }
    // running state of the episode
    if (hipMemsetAsync(ws + a.lo.P, 0, a.lo.counters + sizeof(NsCounters) - a.lo.P, st) != hipSuccess) return DHTS_E_LAUNCH;
    const int nmax = C > L ? C : L;
    ns_init_fwd_kernel<<<(nmax + 255) / 256, 256, 0, st>>>(a);
    const size_t lds_b = sizeof(float) * 2 * (size_t)(Lm > 0 ? Lm : 1), lds_c = sizeof(int) * (size_t)L;
    if (lds_b > 48 * 1024 || lds_c > 48 * 1024) return DHTS_E_INVALID;
    const int ghost_blocks = (2 * L + kNsBlock - 1) / kNsBlock;
    float *ghost = reinterpret_cast<float *>(ws + a.lo.ghost);
    float *tape0 = reinterpret_cast<float *>(ws + a.lo.tape);
    (void)plane;
    for (int step = 0; step < T; ++step) {
        ns_boundary_fwd_kernel<<<1 + ghost_blocks, kNsBlock, lds_b, st>>>(a, step, action);
        const float *cur = hist + (size_t)step * 4 * C;
        float *nxt = hist + (size_t)(step + 1) * 4 * C;
        size_t tape_off = 0;
        for (int g = 0; g < t->n_groups; ++g) {
            const dhts_netstep_group &gr = t->groups[g];
            dhts_macro_desc md = {gr.n_lanes, gr.n_cells, d->dt, gr.dx, d->u_max};
            const size_t c0 = gr.cell0;
            float *tp = hard ? nullptr : tape0 + (size_t)step * a.lo.tape_step + tape_off;
            const int rc = dhts_macro_step_fwd(&md, cur + c0, cur + C + c0, cur + 2 * C + c0, cur + 3 * C + c0, ghost + (size_t)gr.lane_pos0 * 8,
                                               nxt + c0, nxt + C + c0, nxt + 2 * C + c0, nxt + 3 * C + c0, tp, err, stream);
            if (rc != DHTS_OK) return rc;
            tape_off += dhts_macro_step_tape_bytes(&md) / sizeof(float);
        }
        ns_convert_fwd_kernel<<<1, kNsBlock, lds_c, st>>>(a, step);
    }
    ns_reward_kernel<<<1, kNsBlock, sizeof(float) * 2 * (size_t)L, st>>>(a);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

int dhts_netstep_rollout_bwd(const dhts_net_desc *d, const dhts_netstep_tables *t, const float *action, const float *hist,
                             const float *queue, const float *g_reward, float *g_action, void *workspace, dhts_error *err,
                             void *stream) {
    if (!ns_ok(d, t) || !action || !hist || !queue || !g_action || !workspace) return DHTS_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    NsArgs a = ns_args(d, t, 0, const_cast<float *>(hist), const_cast<float *>(queue), nullptr, nullptr, workspace, err);
    const int L = a.L, C = a.C, T = a.T, Lm = a.Lm;
    char *ws = a.ws;
    if (t->persistent) {
        const NsPlan pl = ns_plan(a, ns_al16(sizeof(double) * (2 * (size_t)(Lm > 0 ? Lm : 1) + 8 * (size_t)kNsBlock)), true);
        const size_t lds = (size_t)pl.scratch + pl.tables + pl.st + pl.gl + pl.rows + pl.misc;
        if (lds > 160 * 1024 ||
            hipFuncSetAttribute((const void *)ns_persist_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DHTS_E_INVALID;
        ns_persist_bwd_kernel<<<d->n_replicas, kNsBlock, lds, st>>>(a, action, g_reward, g_action, pl);
        return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
    }
    // cotangents start at zero; the lanes hold what the forward left (lane_n)
    if (hipMemsetAsync(ws + a.lo.G, 0, a.lo.n_bwd - a.lo.G, st) != hipSuccess) return DHTS_E_LAUNCH;
    if (Lm > 0 && hipMemcpyAsync(ws + a.lo.n_bwd, ws + a.lo.lane_n, sizeof(int) * (size_t)Lm, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return DHTS_E_LAUNCH;
    const size_t lds_m = sizeof(double) * (2 * (size_t)(Lm > 0 ? Lm : 1) + 8 * (size_t)kNsBlock);
    if (lds_m > 160 * 1024 ||
        hipFuncSetAttribute((const void *)ns_micro_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m) != hipSuccess)
        return DHTS_E_INVALID;
    float *G = reinterpret_cast<float *>(ws + a.lo.G);
    double *g_ghost = reinterpret_cast<double *>(ws + a.lo.g_ghost);
    const float *tape0 = reinterpret_cast<const float *>(ws + a.lo.tape);
    const int m = L > a.sq ? L : a.sq;
    for (int step = T - 1; step >= 0; --step) {
        ns_micro_bwd_kernel<<<1, kNsBlock, lds_m, st>>>(a, step, action, g_reward);
        float *Gn = G + (size_t)((step + 1) & 1) * 3 * C, *Gp = G + (size_t)(step & 1) * 3 * C;
        if (C > 0 && hipMemsetAsync(Gp + 2 * (size_t)C, 0, sizeof(float) * (size_t)C, st) != hipSuccess) return DHTS_E_LAUNCH;
        size_t tape_off = 0;
        for (int g = 0; g < t->n_groups; ++g) {
            const dhts_netstep_group &gr = t->groups[g];
            dhts_macro_desc md = {gr.n_lanes, gr.n_cells, d->dt, gr.dx, d->u_max};
            const size_t c0 = gr.cell0;
            const int rc = dhts_macro_step_bwd(&md, tape0 + (size_t)step * a.lo.tape_step + tape_off, Gn + c0, Gn + C + c0, Gp + c0, Gp + C + c0,
                                               g_ghost + (size_t)gr.lane_pos0 * 4, err, stream);
            if (rc != DHTS_OK) return rc;
            tape_off += dhts_macro_step_tape_bytes(&md) / sizeof(float);
        }
        if (C > 0) {
            ns_ghosts_bwd_kernel<<<(2 * L + 255) / 256, 256, 0, st>>>(a, step, action);
            ns_ghosts_gather_kernel<<<(m + 255) / 256, 256, 0, st>>>(a, step);
        }
    }
    ns_finish_bwd_kernel<<<(a.n_action + 255) / 256, 256, 0, st>>>(a, g_action);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

}  // extern "C"
