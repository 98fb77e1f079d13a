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

#include <type_traits>

#include "../../include/dhts.h"
#include "arz_device.hpp"
#include "idm_device.hpp"
#include "net_device.hpp"

namespace dhts {

#ifndef DHTS_NS_BLOCK
#define DHTS_NS_BLOCK 1024
#endif
constexpr int kNsBlock = DHTS_NS_BLOCK;       // threads of the one-workgroup kernels (a -DDHTS_NS_BLOCK=512 build: tools/build_variants.sh)
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
    int n_max, pad;         // n_max: the most vehicles any micro lane has held so far (bounds the loops over vehicle slots)
};

struct NsLayout {           // byte offsets into the workspace
    size_t ghost, g_ghost, slot, own_hist, tape, P, V, A, vroute, vcur, lane_n, rused, capv, counters, sigS, lossS, hg, idm_tape,
        nv_idm, adm, nv_post, vx, kc_cell, kc_veh, caprec, capflag, ev, ev_off, G, gP, gV, gA, g_cap, g_own, g_act, n_bwd, ghd, pF, pAB,
        ptape, total;
    size_t tape_step;       // floats of macro tape per step
};
__host__ __device__ inline size_t ns_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Where a pointer of the argument block points: the stepwise kernels and the un-staged persistent form read everything through generic
// pointers into HBM / L2; the persistent kernels that stage the static tables, the per-step rows, ghosts / slots (TB) and the state
// rows / cotangent planes (ST) in LDS hold them as address_space(3) pointers, so that the SAME device functions compile to ds_read /
// ds_write there (a flat load of an LDS address costs a dependent look-up ~700 cycles, a ds_read ~100: section 9 of DESIGN.md).
#if defined(__HIP_DEVICE_COMPILE__)
#define NS_LDS __attribute__((address_space(3)))
#define NS_GLOBAL __attribute__((address_space(1)))
#else                       // (the host pass only parses the device functions: there the qualified types do not convert to generic ones)
#define NS_LDS
#define NS_GLOBAL
#endif
template <int AS, class T> struct NsPtrT { typedef T *type; };                     // 0: generic (what the host passes)
template <class T> struct NsPtrT<1, T> { typedef NS_GLOBAL T *type; };             // 1: global memory
template <class T> struct NsPtrT<3, T> { typedef NS_LDS T *type; };                // 3: LDS

struct NsCommon {
    int L, C, T, sq, F, n_action, Lm, cap, ncap, hard, route_stride, n_draws, max_events, loss_steps;
    float um, dtf, vlen, s0f;
    double um_d, dt_d, vlen_d;
    const int32_t *left_src, *left_gate, *right_src, *conv_next;     // [T][L] per-step tables (per replica: table_stride)
    const double *schedule;
    const int32_t *routes, *route_ptr;
    const double *draws;
    int NI, n_edges, n_islots, has_source, n_routes, tensor_ladder;
    const double *veh_params;        // dhts_hybrid_tables::veh_params: [n_routes][6] beside the route table, or NULL (default vehicles)
    long long table_stride, draws_stride;    // elements between replicas in the [T][L] tables / the draws (0 = shared)
    char *ws;
    NsLayout lo;
    float *hist, *queue, *reward;
    int32_t *counts;
    dhts_error *err;
};
// the static tables: X(name, element type, count expression over (a))
#define NS_TABLES(X) \
    X(lane_ncell, int32_t, a.L) X(lane_off, int32_t, a.L) X(sig_kind, int32_t, a.L) X(inter, int32_t, a.L) X(lane_macro, int32_t, a.L) \
    X(lane_gpos, int32_t, a.L) X(lane_mslot, int32_t, a.L) X(lane_cslot, int32_t, a.L) X(lane_source, int32_t, (a.has_source ? a.L : 0)) \
    X(lane_dx, double, a.L) X(lane_len, double, a.L) X(cell_lane, int32_t, a.C) X(micro_lanes, int32_t, a.Lm) X(cap_lanes, int32_t, a.ncap) \
    X(nxt_ptr, int32_t, a.L + 1) X(prv_ptr, int32_t, a.L + 1) X(nxt_idx, int32_t, a.n_edges) X(prv_idx, int32_t, a.n_edges) \
    X(inter_ptr, int32_t, a.sq + 1) X(inter_idx, int32_t, a.n_islots)

// the running state of the micro side (forward: vehicles, lane counts, capacitors, counters; reverse: their cotangents), in the
// order of the workspace layout: X(name, element type, count over (a), planes)
#define NS_MICRO_FWD(X) \
    X(P, float, 2 * plane) X(V, float, 2 * plane) X(A, float, plane) X(vroute, int, plane) X(vcur, int, plane) X(lane_n, int, a.Lm) \
    X(rused, int, a.L) X(capv, float, a.ncap) X(counters, NsCounters, 1)
#define NS_MICRO_BWD(X) \
    X(gP, double, 2 * plane) X(gV, double, 2 * plane) X(gA, double, plane) X(g_cap, double, a.ncap) X(g_own, float, 2 * a.L) \
    X(g_act, double, a.n_action) X(n_bwd, int, a.Lm)
// what only a persistent kernel's copy of the argument block holds (LDS pointers are 4 bytes on the device and do not exist on the host:
// the block the host builds and the kernels receive, NsArgs, must not contain any)
template <bool kDevice> struct NsStaged {};
template <> struct NsStaged<true> {
#define NS_X(name, ty, count) NS_LDS ty *m_##name;
    NS_MICRO_FWD(NS_X) NS_MICRO_BWD(NS_X)
#undef NS_X
    // the per-step table rows of steps t and t + 1 ([2][L] each), ghosts / slots / ghost cotangents, the signal table (TB); the state rows
    // t and t + 1 ([2][4][C], row r in copy r & 1) and the cotangent planes ([2][3][C]) (ST)
    NS_LDS float *gh, *sl, *sg, *ow;        // ow [2][2 L]: the lanes' stored downstream ghosts before steps t and t + 1 (forward)
    NS_LDS int32_t *sgi;
    NS_LDS double *gg;
    NS_LDS int32_t *row_i;       // [2][4][L]: left_src, left_gate, right_src, conv_next
    NS_LDS double *row_d;        // [2][L]: schedule
    NS_LDS float *st, *gl;
    NS_LDS const int32_t *routes_l, *rlen_l;     // the route table and its rows' lengths (null: not staged)
    NS_LDS const double *lane_cc, *lane_cfl;     // dt / dx and dx / dt of every lane (TB: a cell item would divide twice for them)
};
template <int TA, int SS, bool MS = false> struct NsArgsT : NsCommon, NsStaged<TA != 0> {    // TA: address space of the static tables (3 = staged, with rows and ghosts)
    static constexpr bool kTB = TA == 3, kST = SS == 2, kSG = SS >= 1, kMS = MS;    // SS: 2 = state rows and cotangent planes in LDS, 1 = the planes only
#define NS_X(name, ty, count) typename NsPtrT<TA, const ty>::type name;
    NS_TABLES(NS_X)
#undef NS_X
};
typedef NsArgsT<0, 0> NsArgs;              // what the host builds and every kernel receives

// (the persistent kernels read the argument block from LDS: a pointer loaded from there is generic to the compiler -- flat_load, which also
// counts against the LDS wait counter -- unless its TYPE says that it points into global memory)
template <typename T> __device__ __forceinline__ NS_GLOBAL T *ns_glob(T *p) { return (NS_GLOBAL T *)p; }
template <typename T> __device__ __forceinline__ NS_GLOBAL T *ns_ptr(const NsCommon &a, size_t off) {
    return (NS_GLOBAL T *)reinterpret_cast<T *>(a.ws + off);
}
// an array of the micro side's running state: in LDS (MS) or in the workspace
#define NS_MS(a, T, X) ([&]() { if constexpr (std::decay_t<decltype(a)>::kMS) return (a).m_##X; else return ns_ptr<T>(a, (a).lo.X); }())
// the state before / after step t, the cotangent planes of the state after step r, the per-step table rows of step t
template <class A> __device__ __forceinline__ auto ns_state(const A &a, int r) {
    if constexpr (A::kST) return a.st + (size_t)(r & 1) * 4 * a.C;
    else return ns_glob(a.hist) + (size_t)r * 4 * a.C;
}
template <class A> __device__ __forceinline__ auto ns_G(const A &a, int r) {
    if constexpr (A::kSG) return a.gl + (size_t)(r & 1) * 3 * a.C;
    else return ns_ptr<float>(a, a.lo.G) + (size_t)(r & 1) * 3 * a.C;
}
template <bool TB> struct NsRowT {
    typename NsPtrT<TB ? 3 : 1, const int32_t>::type left_src, left_gate, right_src, conv_next;
    typename NsPtrT<TB ? 3 : 1, const double>::type schedule;
};
template <class A> __device__ __forceinline__ NsRowT<A::kTB> ns_row(const A &a, int t) {
    NsRowT<A::kTB> r;
    if constexpr (A::kTB) {
        const auto b = a.row_i + (size_t)(t & 1) * 4 * a.L;
        r.left_src = b; r.left_gate = b + a.L; r.right_src = b + 2 * a.L; r.conv_next = b + 3 * a.L;
        r.schedule = a.row_d + (size_t)(t & 1) * a.L;
    } else {
        const size_t o = (size_t)t * a.L;
        r.left_src = ns_glob(a.left_src) + o; r.left_gate = ns_glob(a.left_gate) + o; r.right_src = ns_glob(a.right_src) + o;
        r.conv_next = ns_glob(a.conv_next) + o; r.schedule = ns_glob(a.schedule) + o;
    }
    return r;
}
template <class A> __device__ __forceinline__ auto ns_ghosts(const A &a) {
    if constexpr (A::kTB) return a.gh; else return ns_ptr<float>(a, a.lo.ghost);
}
template <class A> __device__ __forceinline__ auto ns_slots(const A &a) {
    if constexpr (A::kTB) return a.sl; else return ns_ptr<float>(a, a.lo.slot);
}
template <class A> __device__ __forceinline__ auto ns_gghost(const A &a) {
    if constexpr (A::kTB) return a.gg; else return ns_ptr<double>(a, a.lo.g_ghost);
}

// inclusive block scan (blockDim.x = multiple of 64, <= 1024); `w` = 16 elements of LDS scratch; returns the inclusive prefix,
// `total` = the block's sum.  Two barriers; the wavefronts' sums are scanned by every wavefront's lanes 0..15 (one LDS read + a wave
// scan instead of a 16-step serial loop: 2 000 cycles a call in round 5's stamps).
__device__ __forceinline__ int ns_readlane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ double ns_readlane(double x, int l) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ void ns_wave_sums(const T *w, T &off, T &total) {      // w[k] = sum of wavefront k; off = sum of the wavefronts before mine
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const T v = wave_scan_add(lane < nw ? w[lane] : (T)0);
    total = ns_readlane(v, 63);
    const T before = ns_readlane(v, wv > 0 ? wv - 1 : 0);
    off = wv > 0 ? before : (T)0;
}
template <typename T>
__device__ __forceinline__ T ns_block_scan(T x, T *w, T &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    T s = wave_scan_add(x);
    if (lane == 63) w[wv] = s;
    __syncthreads();
    T off;
    ns_wave_sums(w, off, total);
    __syncthreads();
    return s + off;
}
// two scans behind one pair of barriers
template <typename T1, typename T2>
__device__ __forceinline__ void ns_block_scan2(T1 x1, T2 x2, T1 *w1, T2 *w2, T1 &incl1, T2 &incl2, T1 &total1, T2 &total2) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const T1 s1 = wave_scan_add(x1);
    const T2 s2 = wave_scan_add(x2);
    if (lane == 63) { w1[wv] = s1; w2[wv] = s2; }
    __syncthreads();
    T1 off1; T2 off2;
    ns_wave_sums(w1, off1, total1);
    ns_wave_sums(w2, off2, total2);
    __syncthreads();
    incl1 = s1 + off1; incl2 = s2 + off2;
}

// The vehicle slots worth visiting: slot i < nq of every micro lane, nq = the smallest power of two >= the fullest lane so far (<= cap).
// Item j -> (micro lane m, slot i, index into the [Lm][cap] arrays).
struct NsSlots { int nq, sh, total, cap; };
__device__ __forceinline__ NsSlots ns_slot_range(int Lm, int cap, int n_max) {
    NsSlots r; r.cap = cap;
    int sh = 0;
    while ((1 << sh) < n_max) sh++;
    if ((1 << sh) >= cap) { r.nq = cap; r.sh = -1; } else { r.nq = 1 << sh; r.sh = sh; }
    r.total = Lm * r.nq;
    return r;
}
__device__ __forceinline__ void ns_slot_of(const NsSlots &r, int j, int &m, int &i, size_t &idx) {
    m = r.sh >= 0 ? j >> r.sh : j / r.cap;
    i = j - m * r.nq;
    idx = (size_t)m * r.cap + i;
}

// sub-phase stamps of thread 0 (a -DDHTS_NS_STAMPS build; tools/probes/exp_persist_stamps.py)
#ifdef DHTS_NS_STAMPS
__shared__ long long ns_sub_[24], ns_sub_last_;
#define NS_SUB0() { if (threadIdx.x == 0) ns_sub_last_ = __builtin_amdgcn_s_memtime(); }
#define NS_SUB(i) { if (threadIdx.x == 0) { const long long n_ = __builtin_amdgcn_s_memtime(); ns_sub_[i] += n_ - ns_sub_last_; ns_sub_last_ = n_; } }
#else
#define NS_SUB0()
#define NS_SUB(i)
#endif

// ---- forward-mode duals of the head gap: value + gradient w.r.t. (p_h, v_h, p_l, v_l, s_prev, s_cur, s_next); every component follows
// the float32 rule of the reference's operator (sum, product, quotient, sigmoid), so the VALUES are the reference's bit for bit.
// (Eight threads per lane with one component each were tried in round 5: no faster at 28 lanes -- the phase's cost were its scans --
// and two passes instead of one at 144.)
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
__device__ __forceinline__ D7 d7_soft(D7 a, float k, bool exact = false) {          // dmath.operation.sigmoid(value, constant = k)
    float s, ds;
    soft_switch_both(a.v, k, s, ds, exact);
    D7 x; x.v = s;
#pragma unroll
    for (int i = 0; i < 7; ++i) x.g[i] = a.g[i] * ds;
    return x; }
__device__ __forceinline__ D7 d7_pos_or_zero(D7 a) { return a.v > 0.f ? a : d7_c(0.f); }     // x if x > 0 else 0.0

// The signals of a step: every ghost, head gap and their adjoints need the green of some lane = one of the two switches of its
// intersection (phase_signal_at: divisions, two sigmoids).  The staging persistent kernels evaluate them ONCE per intersection and step
// into a table in LDS (sg[t & 1][sq][4] = we, ns, d we / d action, d ns / d action; sgi[t & 1] = the action index of intersection 0), one step
// ahead; everybody else evaluates them in place -- the same float expressions either way.
template <class A> __device__ __forceinline__ void ns_signal_fill(const A &a, const float *action, int t, int q) {
    if constexpr (A::kTB) {
        if (q >= a.sq) return;
        float we, ns, av, pr; int ai;
        const bool hard = a.hard != 0;
        phase_signal_at(action, a.n_action, a.sq, a.F, t / a.F, t % a.F, q, we, ns, av, pr, ai, hard, a.tensor_ladder != 0);
        const float z = (av - pr) * kSigK;
        const bool sat = hard || z < -16.f || z > 16.f;
        const auto s4 = a.sg + ((size_t)(t & 1) * a.sq + q) * 4;
        s4[0] = we; s4[1] = ns; s4[2] = sat ? 0.f : we * (1.f - we) * kSigK; s4[3] = sat ? 0.f : -(ns * (1.f - ns) * kSigK);
        if (q == 0) a.sgi[t & 1] = ai;
    }
}
template <class A> __device__ __forceinline__ float ns_lane_signal(const A &a, const float *action, int t, int lid, bool hard, float *ds_da, int *a_index) {
    const int kd = a.sig_kind[lid];
    if (kd == 0) { if (ds_da) *ds_da = 0.f; if (a_index) *a_index = -1; return 1.f; }
    if constexpr (A::kTB) {
        const int k = a.inter[lid];
        const auto s4 = a.sg + ((size_t)(t & 1) * a.sq + k) * 4;
        if (ds_da) *ds_da = kd == 1 ? s4[2] : s4[3];
        if (a_index) *a_index = a.sgi[t & 1] + k;
        return kd == 1 ? s4[0] : s4[1];
    } else {
        float we, ns, av, pr; int ai;
        phase_signal_at(action, a.n_action, a.sq, a.F, t / a.F, t % a.F, a.inter[lid], we, ns, av, pr, ai, hard, a.tensor_ladder != 0);
        if (ds_da) {
            const float z = (av - pr) * kSigK;
            const bool sat = hard || z < -16.f || z > 16.f;
            *ds_da = sat ? 0.f : (kd == 1 ? we * (1.f - we) * kSigK : -(ns * (1.f - ns) * kSigK));
        }
        if (a_index) *a_index = ai;
        return kd == 1 ? we : ns;
    }
}

// entry k of route row `row` / the row's length (entries before the -1 padding): LDS copies where a persistent kernel staged them
template <class A> __device__ __forceinline__ int ns_route_at(const A &a, int row, int k) {
    if constexpr (A::kTB) { if (a.routes_l != nullptr) return a.routes_l[(size_t)row * a.route_stride + k]; }
    return ns_glob(a.routes)[(size_t)row * a.route_stride + k];
}
template <class A> __device__ __forceinline__ int ns_route_len(const A &a, int row) {
    if constexpr (A::kTB) { if (a.rlen_l != nullptr) return a.rlen_l[row]; }
    const auto route = ns_glob(a.routes) + (size_t)row * a.route_stride;
    int rlen = 0;
    while (rlen < a.route_stride && route[rlen] >= 0) rlen++;
    return rlen;
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
// items: side 0 of all lanes, then -- from the next multiple of 64 -- side 1 (a wavefront runs ONE of the two branches)
__host__ __device__ inline int ns_ghost_items(int L) { return ((L + 63) & ~63) + L; }
template <class A> __device__ __forceinline__ void ns_ghost_fwd_item(const A &a, int t, const float *__restrict__ action, int item) {
    const int L = a.L, C = a.C;
    const bool hard = a.hard != 0;
    const float um = a.um;
    const auto cur = ns_state(a, t);
    const auto rw = ns_row(a, t);
    {
        const int Lp = (L + 63) & ~63;
        const int side = item >= Lp ? 1 : 0, lane = item - side * Lp;
        if (lane >= L) return;
        auto own_out = ns_ptr<float>(a, a.lo.own_hist) + (size_t)(t + 1) * 2 * L;       // (the history the reverse sweep reads)
        float own_r = 0.f, own_u = 0.f;
        if (side == 1) {
            if constexpr (A::kTB) { const auto o = a.ow + (size_t)(t & 1) * 2 * L; own_r = o[2 * lane]; own_u = o[2 * lane + 1]; }
            else { const auto o = ns_ptr<float>(a, a.lo.own_hist) + (size_t)t * 2 * L; own_r = o[2 * lane]; own_u = o[2 * lane + 1]; }
        }
        if (!a.lane_macro[lane]) {
            if (side == 1) {
                own_out[2 * lane] = own_r; own_out[2 * lane + 1] = own_u;
                if constexpr (A::kTB) { const auto o = a.ow + (size_t)((t + 1) & 1) * 2 * L; o[2 * lane] = own_r; o[2 * lane + 1] = own_u; }
            }
            return;
        }
        float fr, fu, fy, fq;
        if (side == 0) {
            const int ls = rw.left_src[lane], lg = rw.left_gate[lane];
            if (ls == -1) {                    // source lane: Python floats in the reference, read as such by its solve (ghost_source_pack)
                ghost_source_pack(rw.schedule[lane], fr, fy, fu, fq);
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
            float gr = own_r, gu = own_u;
            if (rs >= 0) { const int first = a.lane_off[rs]; gr = cur[first]; gu = cur[2 * C + first]; }
            const float sg = ns_lane_signal(a, action, t, lane, hard, nullptr, nullptr);
            const float s2 = hard ? (sg > 0.5f ? 1.f : 0.f) : soft_switch(sg - 0.5f, kSigK);
            fr = s2 * gr + (1.0f - s2) * 1.0f;
            fu = s2 * gu + (1.0f - s2) * 0.0f;
            glue_from_r_u(fr, fu, um, fy, fq);
            own_out[2 * lane] = fr; own_out[2 * lane + 1] = fu;
            if constexpr (A::kTB) { const auto o = a.ow + (size_t)((t + 1) & 1) * 2 * L; o[2 * lane] = fr; o[2 * lane + 1] = fu; }
        }
        const auto g = ns_ghosts(a) + ((size_t)a.lane_gpos[lane] * 2 + side) * 4;
        g[0] = fr; g[1] = fy; g[2] = fu; g[3] = fq;
    }
}

// the micro side of a step's boundary + the IDM steps: ONE workgroup (all its threads call; hd_s = [Lm][2] floats of LDS)
template <class A> __device__ __forceinline__ void ns_micro_fwd(const A &a, int t, const float *__restrict__ action, float *hd_s) {
    const int tid = threadIdx.x, B = blockDim.x;
    const bool hard = a.hard != 0;
    const int Lm = a.Lm, cap = a.cap;
    if (Lm == 0) return;
    __shared__ double scan_d[16];
    __shared__ int scan_i[16];
    const size_t plane = (size_t)Lm * cap;
    auto P0 = NS_MS(a, float, P) + (size_t)(t & 1) * plane, P1 = NS_MS(a, float, P) + (size_t)((t + 1) & 1) * plane;
    auto V0 = NS_MS(a, float, V) + (size_t)(t & 1) * plane, V1 = NS_MS(a, float, V) + (size_t)((t + 1) & 1) * plane;
    auto A_ = NS_MS(a, float, A);
    auto vroute = NS_MS(a, int, vroute), vcur = NS_MS(a, int, vcur), lane_n = NS_MS(a, int, lane_n);
    auto rused = NS_MS(a, int, rused);
    auto cnt = NS_MS(a, NsCounters, counters);
    auto adm = hard ? nullptr : ns_ptr<int>(a, a.lo.adm) + (size_t)t * Lm;
    const float vlen = a.vlen;

    NS_SUB0()
    // ---- admission on micro source lanes (_simulator.py:153-174): room for half a vehicle at the entrance, then ONE draw of the
    // host's stream per such lane, lanes in id order ----
    if (a.has_source) {
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
                    const double draw = ns_glob(a.draws)[idx];
                    const int r_lo = ns_glob(a.route_ptr)[l], r_n = ns_glob(a.route_ptr)[l + 1] - r_lo;
                    if (draw < ns_row(a, t).schedule[l] && rused[l] < r_n) {
                        if (n >= cap) net_fault(a.err, DHTS_FAULT_CAPACITY, t, l, n);
                        else {
                            const size_t b = (size_t)m * cap;
                            ns_shift_in(P0, V0, A_, vroute, vcur, b, n);
                            P0[b] = 0.f; V0[b] = 0.f; A_[b] = vlen; vroute[b] = r_lo + rused[l]; vcur[b] = 0;
                            rused[l] += 1; lane_n[m] = n + 1;
                            atomicMax(&cnt->n_max, n + 1);
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
        __syncthreads();
    } else if (adm) {
        for (int m = tid; m < Lm; m += B) adm[m] = 0;      // (nothing the head gaps read: no barrier -- in the persistent kernel the wavefronts
    }                                                      //  without a ghost item start on the head gaps while the others blend ghosts)

    NS_SUB(0)
    // ---- head gaps (lane-id order for the running mean of the final signal) ----
    auto hgrow = hard ? nullptr : ns_ptr<NsHeadGap>(a, a.lo.hg) + (size_t)t * Lm;
    auto sigS = ns_ptr<double>(a, a.lo.sigS);
    long long sig_n0 = cnt->sig_n;                     // (written again behind this phase's scans: their barriers order it)
    double sig_sum0 = cnt->sig_sum;
    // the items sit at the END of the workgroup when they fit one pass (the first wavefronts hold the ghost items); thread order = lane order
    const int item0 = Lm <= B ? B - ((Lm + 63) & ~63) : 0;
    for (int base = 0; base < Lm; base += B) {
        const int item = base + tid - item0;
        const int m = item >= 0 ? item : Lm;
        bool occ = false;
        D7 green_dp = d7_c(1000.f), green_dv = d7_c(0.f), red_dp = d7_c(0.f), fin = d7_c(0.f);
        int leader = -1, sgl[3] = {-1, -1, -1};
        if (m < Lm) {
            const int n = lane_n[m];
            if (n > 0) {
                occ = true;
                const int l = a.micro_lanes[m];
                const size_t hs = (size_t)m * cap + n - 1;
                const bool sig_exact = a.tensor_ladder != 0;      // (net_device.hpp sig_exp: `micro` mode takes the exponential rounded once)
                const D7 hp = d7_var(P0[hs], 0), hv = d7_var(V0[hs], 1);
                const int rrow = vroute[hs];
                const int cursor = vcur[hs];
                const int rlen = ns_route_len(a, rrow);
                const D7 Lc = d7_c((float)a.lane_len[l]);
                // leader further along the route (road_network.py:459-580)
                D7 reach = d7_sub(d7_sub(Lc, hp), d7_c(vlen * 0.5f));
                for (int k = cursor; k < rlen - 1; k++) {
                    const int there = ns_route_at(a, rrow, k + 1);
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
                if (prev_exist && !hard) prev_s = d7_soft(d7_sub(d7_c(0.f), hp), 16.f, sig_exact);
                const D7 curr_s = hard ? d7_c(1.f) : d7_mul(d7_soft(hp, 16.f, sig_exact), d7_soft(d7_sub(Lc, hp), 16.f, sig_exact));
                if (next_exist && !hard) next_s = d7_soft(d7_sub(hp, Lc), 16.f, sig_exact);
                const D7 total = d7_add(d7_add(prev_s, curr_s), next_s);
                for (int w = 0; w < 3; w++) {
                    if ((w == 0 && !prev_exist) || (w == 2 && !next_exist)) continue;
                    const int lid = ns_route_at(a, rrow, cursor + w - 1);
                    const D7 sc = w == 0 ? prev_s : (w == 1 ? curr_s : next_s);
                    D7 sv;
                    if (a.sig_kind[lid] == 0) sv = d7_c(1.f);
                    else { sv = d7_var(ns_lane_signal(a, action, t, lid, hard, nullptr, nullptr), 4 + w); sgl[w] = lid; }
                    fin = d7_add(fin, d7_mul(d7_div(sc, total), sv));
                }
            }
        }
        NS_SUB(16)
        // ordered running mean of the final signal over the occupied lanes (signal_rms: _simulator.py:234-262; rms.py)
        float hdp = 1000.f, hdv = 0.f;
        if (hard) {
            if (occ) { const bool green = fin.v >= 0.5f; hdp = green ? green_dp.v : red_dp.v; hdv = green ? green_dv.v : 0.f; }
        } else {
            int tot_n; double tot_s;
            int rank; double incl;
            ns_block_scan2<int, double>(occ ? 1 : 0, occ ? (double)fin.v : 0., scan_i, scan_d, rank, incl, tot_n, tot_s);
            NS_SUB(17)
            if (occ) {
                const long long i = sig_n0 + rank - 1;               // 0-based index of this sample in the stream
                const double S = sig_sum0 + incl;
                sigS[i] = S;
                const double mean = (i + 1 > kNsWindow) ? (S - sigS[i - kNsWindow]) / (double)kNsWindow : S / (double)(i + 1);
                const float k2 = 32.f / fabsf((float)mean);
                const D7 fs = d7_soft(d7_sub(fin, d7_c(0.5f)), k2, a.tensor_ladder != 0);
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

    NS_SUB(1)
    // ---- one IDM step of every vehicle (explicit Euler; _micro_lane.py:131-214, _idm.py:6-50, didm.py:13-103) ----
    IdmParams prm;              // MicroVehicle.default_micro_vehicle(speed_limit), micro_vehicle.py:31-72
    prm.a_max = a.um_d * 1.0; prm.a_pref = a.um_d * 0.8; prm.v_target = a.um_d * 0.9; prm.min_space = a.vlen_d * 0.1; prm.time_pref = 0.1;
    prm.length = a.vlen_d;
    auto tape = hard ? nullptr : ns_ptr<float4>(a, a.lo.idm_tape) + (size_t)t * plane;
    auto nv_idm = hard ? nullptr : ns_ptr<int>(a, a.lo.nv_idm) + (size_t)t * Lm;
    const NsSlots sr = ns_slot_range(Lm, cap, cnt->n_max);
    const bool f32_ladder = a.tensor_ladder && !hard;        // (dhts_hybrid_tables::micro_tensor_ladder: said by the host, not inferred from the source lanes)
    for (int j = tid; j < sr.total; j += B) {
        int m, i; size_t idx;
        ns_slot_of(sr, j, m, i, idx);
        const int n = lane_n[m];
        if (i == 0 && nv_idm) nv_idm[m] = n;
        if (i >= n) continue;
        IdmStep o;
        auto one_step = [&](const IdmParams &pm) {
            if (f32_ladder) {            // `micro` mode, differentiable: the reference steps these lanes in float32 tensor arithmetic (idm_device.hpp)
                const float p = P0[idx], v = V0[idx];
                float dp, dv, sg = 1.f;
                if (i == n - 1) { dp = hd_s[2 * m]; dv = hd_s[2 * m + 1]; }
                else {
                    const float pl = P0[idx + 1];
                    dp = fabsf(pl - p) - (float)((pm.length + pm.length) * 0.5); dv = v - V0[idx + 1];
                    sg = pl > p ? 1.f : (pl < p ? -1.f : 0.f);
                }
                idm_step_f32(p, v, dp, dv, pm, a.dt_d, o, sg);
            } else {
                const double p = P0[idx], v = V0[idx];
                double dp, dv;
                if (i == n - 1) { dp = (double)hd_s[2 * m]; dv = (double)hd_s[2 * m + 1]; }
                else { dp = fabs((double)P0[idx + 1] - p) - ((pm.length + pm.length) * 0.5); dv = v - (double)V0[idx + 1]; }
                // (differentiable episode: the head gap is a float32 tensor in the reference, its vehicle's step mixed arithmetic -- idm_device.hpp)
                idm_step_lane(P0[idx], V0[idx], dp, dv, i == n - 1 && !hard, pm, a.dt_d, o);
            }
        };
        if (a.veh_params) {          // this vehicle's own attributes: the row of its route (micro_vehicle.py:5-28)
            const double *q = a.veh_params + 6 * (size_t)vroute[idx];
            IdmParams pv;
            pv.a_max = q[0]; pv.a_pref = q[1]; pv.v_target = q[2]; pv.min_space = q[3]; pv.time_pref = q[4]; pv.length = prm.length;
            one_step(pv);
        } else one_step(prm);
        if (o.collided) net_fault(a.err, DHTS_FAULT_COLLISION, t, a.micro_lanes[m], i);
        P1[idx] = o.np; V1[idx] = o.nv;
        if (tape) tape[idx] = make_float4(o.dE[2], o.dE[3], o.dLd[2], o.dLd[3]);
    }
    NS_SUB(2)
}

__global__ void __launch_bounds__(kNsBlock) ns_boundary_fwd_kernel(NsArgs a, int t, const float *__restrict__ action) {
    extern __shared__ float hd_dyn[];                 // [Lm][2] head gaps of this step
    if (blockIdx.x > 0) ns_ghost_fwd_item(a, t, action, (blockIdx.x - 1) * blockDim.x + threadIdx.x);
    else ns_micro_fwd(a, t, action, hd_dyn);
}

// =====================================================================================================================
// forward, part 2: hand-offs in lane-id order on the committed state, then the queue loss
// =====================================================================================================================
template <class A> __device__ __forceinline__ void ns_push_event(const A &a, NsCounters *cnt, int t, const NsEvent &e) {
    if (a.hard) return;
    const int k = cnt->n_events;
    if (k >= a.max_events) { net_fault(a.err, DHTS_FAULT_CAPACITY, t, e.lane, -2); return; }
    ns_ptr<NsEvent>(a, a.lo.ev)[k] = e;
    cnt->n_events = k + 1;
}

template <class A> __device__ __forceinline__ void ns_convert(const A &a, int t, int *cand) {      // cand: [L] ints of LDS (candidate flags of the event walk)
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a.L, C = a.C, Lm = a.Lm, cap = a.cap;
    const bool hard = a.hard != 0;
    const float um = a.um, vlen = a.vlen, dtf = a.dtf, s0f = a.s0f;
    __shared__ double scan_d[16];
    __shared__ int scan_i[16];
    const auto nxt = ns_state(a, t + 1);               // committed state: the lanes' steps wrote it
    const auto rw = ns_row(a, t);
    const auto Rn = nxt, Yn = nxt + C, Un = nxt + 2 * C;
    const size_t row = (size_t)t * L;
    const size_t plane = (size_t)Lm * cap;
    auto P1 = Lm ? NS_MS(a, float, P) + (size_t)((t + 1) & 1) * plane : nullptr;
    auto V1 = Lm ? NS_MS(a, float, V) + (size_t)((t + 1) & 1) * plane : nullptr;
    auto A_ = NS_MS(a, float, A);
    auto vroute = NS_MS(a, int, vroute), vcur = NS_MS(a, int, vcur), lane_n = NS_MS(a, int, lane_n);
    auto rused = NS_MS(a, int, rused);
    auto capv = NS_MS(a, float, capv);
    auto cnt = NS_MS(a, NsCounters, counters);
    auto caprec = ns_ptr<float4>(a, a.lo.caprec) + (hard ? 0 : (size_t)t * a.ncap);
    auto capflag = ns_ptr<int>(a, a.lo.capflag) + (hard ? 0 : (size_t)t * a.ncap);
    auto ev_off = ns_ptr<int>(a, a.lo.ev_off);

    if (Lm > 0) {
        for (int l = tid; l < L; l += B) cand[l] = 0;
        __syncthreads();
        NS_SUB0()
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
        NS_SUB(3)
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
                                const int r_lo = ns_glob(a.route_ptr)[m], r_n = ns_glob(a.route_ptr)[m + 1] - r_lo;
                                if (r_n <= 0 || n >= cap) net_fault(a.err, DHTS_FAULT_CAPACITY, t, m, n);
                                else {
                                    const int rrow = r_lo + rused[m] % r_n;
                                    rused[m] += 1;
                                    ns_shift_in(P1, V1, A_, vroute, vcur, b, n);
                                    P1[b] = 0.f; V1[b] = caprec[k].z;
                                    A_[b] = level - (float)((double)level - (double)vlen);
                                    vroute[b] = rrow; vcur[b] = 0;
                                    lane_n[ms] = n + 1;
                                    if (n + 1 > cnt->n_max) cnt->n_max = n + 1;
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
                                const int cursor = vcur[hs];
                                const int rlen = ns_route_len(a, vroute[hs]);
                                const int nid = cursor < rlen - 1 ? ns_route_at(a, vroute[hs], cursor + 1) : -1;
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
                                            if (n2 + 1 > cnt->n_max) cnt->n_max = n2 + 1;
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
                                        if constexpr (A::kST) {
                                            const auto hn = ns_glob(a.hist) + (size_t)(t + 1) * 4 * C;
                                            hn[cell] = n_r; hn[C + cell] = yy; hn[2 * C + cell] = hv;
                                        }
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

    NS_SUB(4)
    // ---- queue-length loss of the committed state (lane-id order; _env.py:664-742 with :586-618) ----
    auto nv_post = (Lm && !hard) ? ns_ptr<int>(a, a.lo.nv_post) + (size_t)t * Lm : nullptr;
    auto vx = (Lm && !hard) ? ns_ptr<float>(a, a.lo.vx) + (size_t)t * plane : nullptr;
    auto kc_cell = hard ? nullptr : ns_ptr<float>(a, a.lo.kc_cell) + (size_t)t * C;
    auto kc_veh = (Lm && !hard) ? ns_ptr<float>(a, a.lo.kc_veh) + (size_t)t * plane : nullptr;
    auto lossS = ns_ptr<double>(a, a.lo.lossS);
    const long long n_start = cnt->loss_n;              // (written at the end of the previous call, barriers ago; the walk's changes to the state
    long long n0 = n_start;                             //  and the vehicles are behind its own barrier: none needed here)
    double S0 = cnt->loss_sum;
    if (hard) {
        for (int l = tid; l < L; l += B) {
            float qlen = 0.f;
            const bool mac = a.lane_macro[l] != 0;
            if (mac) {
                const int cntl = a.lane_ncell[l], off = a.lane_off[l];
                const float w = (float)a.lane_dx[l];
                for (int i = 0; i < cntl; i++) qlen = qlen + (Un[off + i] < s0f ? 1.f : 0.f) * (Rn[off + i] * w / vlen);
            } else {
                const int ms = a.lane_mslot[l], cntl = lane_n[ms];
                for (int i = 0; i < cntl; i++) qlen = qlen + (V1[(size_t)ms * cap + i] < s0f ? 1.f : 0.f);
            }
            // (an IDM lane counts Python floats there: (n ** 2.0) * dt in double, _env.py:709-738; a cell lane's terms are float32 tensors)
            ns_glob(a.queue)[row + l] = mac ? (qlen * qlen) * dtf : (float)(((double)qlen * (double)qlen) * a.dt_d);
        }
        NS_SUB(5)
        return;
    }
    // The constants of the soft queue indicator come from the running mean of ALL samples so far (cells and vehicles, lanes in id
    // order): (1) a thread per lane counts and sums its samples, an ordered prefix over the lanes follows; (2) a thread per SAMPLE
    // continues the lane's prefix up to its own sample in the same order, takes the mean, the constant and the sample's term;
    // (3) a thread per lane adds its terms in order.  Scratch (behind the candidate flags): laneS [L] | laneN [L] | termC [C] | termV [plane]
    double *laneS = reinterpret_cast<double *>(cand);
    int *laneN = reinterpret_cast<int *>(laneS + L);
    if (Lm == 0) { NS_SUB0() }
    float *termC = reinterpret_cast<float *>(laneN + L), *termV = termC + C;
    for (int base = 0; base < L; base += B) {
        const int l = base + tid;
        int cntl = 0;
        double sum = 0.;
        if (l < L) {
            if (a.lane_macro[l]) {
                cntl = a.lane_ncell[l];
                const int off = a.lane_off[l];
                for (int i = 0; i < cntl; i++) sum += (double)(s0f - Un[off + i]);
            } else {
                const int ms = a.lane_mslot[l];
                cntl = lane_n[ms];
                if (nv_post) nv_post[ms] = cntl;
                for (int i = 0; i < cntl; i++) sum += (double)(s0f - V1[(size_t)ms * cap + i]);
            }
        }
        int tot_n; double tot_s;
        int n_incl; double s_incl;
        ns_block_scan2<int, double>(cntl, sum, scan_i, scan_d, n_incl, s_incl, tot_n, tot_s);
        if (l < L) { laneN[l] = (int)(n0 - n_start) + (n_incl - cntl); laneS[l] = S0 + (s_incl - sum); }
        n0 += tot_n; S0 += tot_s;
    }
    __syncthreads();
    NS_SUB(6)
    for (int c = tid; c < C; c += B) {
        const int l = a.cell_lane[c], off = a.lane_off[l], k = c - off;
        double S = laneS[l];
        for (int i = 0; i < k; i++) S += (double)(s0f - Un[off + i]);
        const float x = s0f - Un[c];
        S += (double)x;
        const long long i_g = n_start + laneN[l] + k;
        lossS[i_g] = S;
        const double mean = (i_g + 1 > kNsWindow) ? (S - lossS[i_g - kNsWindow]) / (double)kNsWindow : S / (double)(i_g + 1);
        const float kk = 16.f / fabsf((float)mean);
        kc_cell[c] = kk;
        termC[c] = soft_switch(x, kk) * (Rn[c] * (float)a.lane_dx[l] / vlen);
    }
    const NsSlots sr = ns_slot_range(Lm, cap, cnt->n_max);
    for (int j = tid; j < sr.total; j += B) {
        int ms, i; size_t idx;
        ns_slot_of(sr, j, ms, i, idx);
        if (i >= lane_n[ms]) continue;
        const int l = a.micro_lanes[ms];
        double S = laneS[l];
        for (int j = 0; j < i; j++) S += (double)(s0f - V1[(size_t)ms * cap + j]);
        const float x = s0f - V1[idx];
        S += (double)x;
        const long long i_g = n_start + laneN[l] + i;
        lossS[i_g] = S;
        const double mean = (i_g + 1 > kNsWindow) ? (S - lossS[i_g - kNsWindow]) / (double)kNsWindow : S / (double)(i_g + 1);
        const float kk = 16.f / fabsf((float)mean);
        kc_veh[idx] = kk; vx[idx] = x;
        termV[idx] = soft_switch(x, kk);
    }
    NS_SUB(7)
    __syncthreads();
    NS_SUB(15)
    for (int l = tid; l < L; l += B) {
        float qlen = 0.f;
        if (a.lane_macro[l]) {
            const int cntl = a.lane_ncell[l], off = a.lane_off[l];
            for (int i = 0; i < cntl; i++) qlen = qlen + termC[off + i];
        } else {
            const int ms = a.lane_mslot[l], cntl = lane_n[ms];
            for (int i = 0; i < cntl; i++) qlen = qlen + termV[(size_t)ms * cap + i];
        }
        ns_glob(a.queue)[row + l] = (qlen * qlen) * dtf;
    }
    NS_SUB(5)
    if (tid == 0) { cnt->loss_n = n0; cnt->loss_sum = S0; }      // (everybody read them before the scans' barriers; the caller's barrier follows)
}

// LDS scratch of ns_convert: the candidate flags of the event walk [L ints], then the loss's lane prefixes and per-sample terms
__host__ __device__ inline size_t ns_convert_scratch(int L, int C, size_t plane) {
    const size_t loss = 12 * (size_t)L + 4 * ((size_t)C + plane) + 16;
    return loss > 4 * (size_t)L ? loss : 4 * (size_t)L;
}

__global__ void __launch_bounds__(kNsBlock) ns_convert_fwd_kernel(NsArgs a, int t) {
    extern __shared__ int cand_dyn[];
    ns_convert(a, t, cand_dyn);
}

// reward = - sum of the queue terms, lanes outermost (ItscpEnv._reward, _env.py:770-797): one thread per lane sums its steps,
// lane 0 of the block the lanes; counts out
template <class A> __device__ __forceinline__ void ns_reward(const A &a, float *part) {          // part: [2][L] floats of LDS
    const int tid = threadIdx.x, B = blockDim.x, L = a.L, T = a.T;
    const int cut = (a.loss_steps > 0 && a.loss_steps < T) ? a.loss_steps : T;
    for (int l = tid; l < L; l += B) {
        float s = 0.f, sc = 0.f;
        for (int t = 0; t < T; t++) { const float q = (-1.0f) * ns_glob(a.queue)[(size_t)t * L + l]; s = s + q; if (t < cut) sc = sc + q; }
        part[l] = s; part[L + l] = sc;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, sc = 0.f;
        for (int l = 0; l < L; l++) { s = s + part[l]; sc = sc + part[L + l]; }
        a.reward[0] = s; a.reward[1] = sc;
        auto cnt = NS_MS(a, NsCounters, counters);
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
// LDS scratch of ns_micro_bwd in bytes: [Lm][2] head-gap cotangents + the compact lists of one pass of lanes, or the loss taps' terms
__host__ __device__ inline int ns_list_room(int Lm) { const int r = (Lm + 63) & ~63; return r < 64 ? 64 : (r > kNsBlock ? kNsBlock : r); }
__host__ __device__ inline size_t ns_micro_bwd_scratch(int L, int C, int Lm, size_t plane) {
    const size_t lists = sizeof(double) * (2 * (size_t)(Lm > 0 ? Lm : 1) + 8 * (size_t)ns_list_room(Lm));
    const size_t taps = sizeof(float) * (3 * (size_t)C + plane + (size_t)L) + 16;
    return lists > taps ? lists : taps;
}
__device__ __forceinline__ void ns_shift_out(double *gP, double *gV, double *gA, size_t base, int n) {       // slot 0 leaves
    for (int i = 0; i + 1 < n; ++i) { gP[base + i] = gP[base + i + 1]; gV[base + i] = gV[base + i + 1]; gA[base + i] = gA[base + i + 1]; }
}

template <class A> __device__ __forceinline__ void ns_micro_bwd(const A &a, int t, const float *__restrict__ action, const float *__restrict__ g_reward,
                                             double *lds_d, int n_max) {     // n_max: the episode's fullest micro lane (NsCounters)      // lds_d: [2 Lm + 8 blockDim] doubles of LDS (head-gap cotangents | compact lists)
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a.L, C = a.C, Lm = a.Lm, cap = a.cap;
    const float um = a.um, vlen = a.vlen, dtf = a.dtf, s0f = a.s0f;
    __shared__ int scan_i[32];
    const auto nxt = ns_state(a, t + 1);
    const auto Rn = nxt, Yn = nxt + C, Un = nxt + 2 * C;
    const auto G = ns_G(a, t + 1);                     // cotangent of (r, y, u) of the state after step t
    const auto Gr = G, Gy = G + C, Gu = G + 2 * C;
    const size_t plane = (size_t)Lm * cap;
    const float grew = g_reward ? g_reward[0] : 1.f;
    const bool taps = a.loss_steps <= 0 || t < a.loss_steps;
    const NsSlots sr = ns_slot_range(Lm, cap, n_max);
    auto gPn = NS_MS(a, double, gP) + (size_t)((t + 1) & 1) * plane, gPp = NS_MS(a, double, gP) + (size_t)(t & 1) * plane;
    auto gVn = NS_MS(a, double, gV) + (size_t)((t + 1) & 1) * plane, gVp = NS_MS(a, double, gV) + (size_t)(t & 1) * plane;
    auto gA = NS_MS(a, double, gA);
    auto g_cap = NS_MS(a, double, g_cap);
    auto n_bwd = NS_MS(a, int, n_bwd);
    auto caprec = ns_ptr<float4>(a, a.lo.caprec) + (size_t)t * a.ncap;
    auto capflag = ns_ptr<int>(a, a.lo.capflag) + (size_t)t * a.ncap;

    NS_SUB0()
    // ---- (a) loss taps: reward = - sum_l q_l^2 dt.  A thread per SAMPLE (cell, vehicle) evaluates its term, a thread per lane adds its
    // lane's terms in order (the queue length) and leaves the lane's seed, a thread per sample applies it.  Scratch: termC [C] | termV
    // [plane] | laneG [L] | the switches and their derivatives [2][C] floats (the head gaps' lists come later in the step) ----
    if (taps) {
        float *termC = reinterpret_cast<float *>(lds_d), *termV = termC + C, *laneG = termV + plane, *swS = laneG + L, *swD = swS + C;
        auto kc_cell = ns_ptr<float>(a, a.lo.kc_cell) + (size_t)t * C;
        auto nv_post = ns_ptr<int>(a, a.lo.nv_post) + (size_t)t * Lm;
        auto vx = ns_ptr<float>(a, a.lo.vx) + (size_t)t * plane;
        auto kv = ns_ptr<float>(a, a.lo.kc_veh) + (size_t)t * plane;
        for (int c = tid; c < C; c += B) {
            const float w = (float)a.lane_dx[a.cell_lane[c]];
            float s_, ds;                      // (the switch and its derivative once: the third pass reads them back)
            soft_switch_both(s0f - Un[c], kc_cell[c], s_, ds);
            swS[c] = s_; swD[c] = ds;
            termC[c] = s_ * (Rn[c] * w / vlen);
        }
        for (int j = tid; j < sr.total; j += B) {
            int ms, i; size_t idx;
            ns_slot_of(sr, j, ms, i, idx);
            if (i < nv_post[ms]) termV[idx] = soft_switch(vx[idx], kv[idx]);
        }
        __syncthreads();
        for (int l = tid; l < L; l += B) {
            float qlen = 0.f;
            if (a.lane_macro[l]) {
                const int n = a.lane_ncell[l], off = a.lane_off[l];
                for (int i = 0; i < n; i++) qlen += termC[off + i];
            } else {
                const int ms = a.lane_mslot[l], n = nv_post[ms];
                for (int i = 0; i < n; i++) qlen = qlen + termV[(size_t)ms * cap + i];
            }
            laneG[l] = (-1.0f * dtf * 2.f * qlen) * grew;
        }
        __syncthreads();
        for (int c = tid; c < C; c += B) {
            const int l = a.cell_lane[c];
            const float w = (float)a.lane_dx[l], g_q = laneG[l];
            const float s_ = swS[c], ds = swD[c];
            Gr[c] += g_q * s_ * (w / vlen);
            Gu[c] += g_q * (Rn[c] * w / vlen) * (-ds);
        }
        for (int j = tid; j < sr.total; j += B) {
            int ms, i; size_t idx;
            ns_slot_of(sr, j, ms, i, idx);
            if (i < nv_post[ms]) gVn[idx] += (double)laneG[a.micro_lanes[ms]] * (double)soft_switch_grad(vx[idx], kv[idx]) * -1.0;
        }
    }
    if (Lm > 0) __syncthreads();       // (without IDM lanes nothing happens between a cell's tap and its fold, both by the cell's own thread)

    if (Lm > 0) {
        NS_SUB(8)
        // ---- (b) the step's hand-off events, newest first (one thread: they are few) ----
        if (tid == 0) {
            auto ev_off = ns_ptr<int>(a, a.lo.ev_off);
            auto ev = ns_ptr<NsEvent>(a, a.lo.ev);
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
        NS_SUB(9)
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
    NS_SUB(10)
    // ---- (c) cells whose speed is u(r, y): fold the speed cotangent into (r, y) (a deposited cell's went to its vehicle above) ----
    for (int c = tid; c < C; c += B) {
        const float gu = Gu[c];
        if (gu != 0.f) { float gr = Gr[c], gy = Gy[c]; glue_u_bwd(Rn[c], Yn[c], um, gu, gr, gy); Gr[c] = gr; Gy[c] = gy; Gu[c] = 0.f; }
    }
    if (Lm == 0) return;
    __syncthreads();

    NS_SUB(11)
    // ---- (d) IDM steps: g[i] = dEgo[i]^T g'[i] + dLeading[i-1]^T g'[i-1]; the head's virtual leader returns to the head and to the gap ----
    double *ghd = lds_d;                               // [Lm][2]
    auto tape = ns_ptr<float4>(a, a.lo.idm_tape) + (size_t)t * plane;
    auto nv_idm = ns_ptr<int>(a, a.lo.nv_idm) + (size_t)t * Lm;
    for (int j = tid; j < sr.total; j += B) {
        int m, i; size_t idx;
        ns_slot_of(sr, j, m, i, idx);
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
    for (int m = tid; m < Lm; m += B) n_bwd[m] = nv_idm[m];      // (equal by construction; keeps a fault from spreading)

    NS_SUB(12)
    // ---- (e) head gaps: Jacobian^T of (head_dp, head_dv) to the head vehicle, to its leader (another lane's tail) and to the signals ----
    auto hgrow = ns_ptr<NsHeadGap>(a, a.lo.hg) + (size_t)t * Lm;
    auto g_act = NS_MS(a, double, g_act);
    const int nB = ns_list_room(Lm);                                        // lanes per pass (<= blockDim.x)
    int *li_key = reinterpret_cast<int *>(lds_d + 2 * (size_t)Lm);          // compact lists: leader items [<= nB], signal items [<= 3 nB]
    double *li_p = lds_d + 2 * (size_t)Lm + (size_t)nB * 2;                 // (keys take 4 nB ints = 2 nB doubles of room)
    double *li_v = li_p + nB;
    double *si_v = li_v + nB;
    int *si_key = li_key + nB;
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
        // (no barrier: lane m's slots are touched by thread m mod blockDim.x alone, here and below)
        // leaders: items compacted in lane order, every target lane adds the ones that name it
        const int has_l = (h.valid && h.leader >= 0) ? 1 : 0;
        int n_sig = 0;
        if (h.valid) for (int w = 0; w < 3; w++) n_sig += h.sig[w] >= 0;
        int tot_l, tot_s;
        int pos_l, pos_s;
        ns_block_scan2<int, int>(has_l, n_sig, scan_i, scan_i + 16, pos_l, pos_s, tot_l, tot_s);
        pos_l -= has_l; pos_s -= n_sig;
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
        if (base + B < Lm) __syncthreads();           // (the lists are reused by the next pass)
    }

    NS_SUB(13)
    // ---- (f) admission undone: the admitted vehicle (position 0, speed 0: constants) leaves through the tail ----
    auto adm = ns_ptr<int>(a, a.lo.adm) + (size_t)t * Lm;
    for (int m = tid; m < Lm; m += B) {
        if (adm[m]) { ns_shift_out(gPp, gVp, gA, (size_t)m * cap, n_bwd[m]); n_bwd[m] -= 1; }
    }
    NS_SUB(14)
}
__global__ void __launch_bounds__(kNsBlock) ns_micro_bwd_kernel(NsArgs a, int t, const float *__restrict__ action, const float *__restrict__ g_reward) {
    extern __shared__ double lds_dyn[];
    ns_micro_bwd(a, t, action, g_reward, lds_dyn, ns_ptr<NsCounters>(a, a.lo.counters)->n_max);
}

// =====================================================================================================================
// reverse, part 2: the lanes' ghost cotangents (dhts_macro_step_bwd's g_ghost) to the neighbours' edge cells, the stored ghosts and
// the action.  slot [L][2][4] = (cotangent for the source cell's r, for its u, the action partial, the action index as a float)
// =====================================================================================================================
template <class A> __device__ __forceinline__ void ns_ghost_bwd_item(const A &a, int t, const float *__restrict__ action, int item) {
    const int L = a.L, C = a.C;
    const int Lp = (L + 63) & ~63;
    const int side = item >= Lp ? 1 : 0, lane = item - side * Lp;
    if (lane >= L) return;
    const int j = 2 * lane + side;
    const auto sl = ns_slots(a) + (size_t)j * 4;
    float add_r = 0.f, add_u = 0.f, a_val = 0.f; int a_key = -1;
    if (a.lane_macro[lane]) {
        const float um = a.um;
        const auto cur = ns_state(a, t);
        const auto rw = ns_row(a, t);
        const auto gg = ns_gghost(a) + ((size_t)a.lane_gpos[lane] * 2 + side) * 2;
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
            auto own_in = ns_ptr<float>(a, a.lo.own_hist) + (size_t)t * 2 * L;
            auto g_own = NS_MS(a, float, g_own);
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
template <class A> __device__ __forceinline__ void ns_ghost_gather_item(const A &a, int t, int m) {
    const int L = a.L, C = a.C;
    const auto slot = ns_slots(a);
    const auto G = ns_G(a, t);
    const auto rw = ns_row(a, t);
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
}
// one WAVEFRONT per intersection q: its (lane, side) slots' action partials summed in a fixed order (inclusive wave scan, 64 slots at a time)
template <class A> __device__ __forceinline__ void ns_action_gather_wave(const A &a, int q) {
    if (q >= a.sq) return;
    const auto slot = ns_slots(a);
    const int lane = threadIdx.x & 63;
    const int lo = a.inter_ptr[q], hi = a.inter_ptr[q + 1];
    double v = 0.; int key = -1;
    for (int base = lo; base < hi; base += 64) {
        const int k = base + lane;
        double x = 0.; int kk = -1;
        if (k < hi) {
            const auto sl = slot + (size_t)a.inter_idx[k] * 4;
            const float key_f = sl[3];
            if (key_f >= 0.f) { x = (double)sl[2]; kk = (int)key_f; }
        }
        v += wave_last(wave_scan_add(x));
        const int km = wave_last(wave_scan_add(kk >= 0 ? 1 : 0));
        if (km) {           // every active slot of an intersection names the same action entry
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(kk >= 0);
            key = __builtin_amdgcn_readlane(kk, (int)__builtin_ctzll(mk));
        }
    }
    if (lane == 0 && key >= 0) NS_MS(a, double, g_act)[key] += v;
}
__global__ void ns_ghosts_gather_kernel(NsArgs a, int t, int lane_blocks) {        // blocks of 256: lane items, then 4 intersections each
    if ((int)blockIdx.x < lane_blocks) ns_ghost_gather_item(a, t, blockIdx.x * blockDim.x + threadIdx.x);
    else ns_action_gather_wave(a, ((int)blockIdx.x - lane_blocks) * 4 + (threadIdx.x >> 6));
}

// =====================================================================================================================
// PERSISTENT form (round 5): the same step -- the very device functions above -- inside ONE kernel per direction, one workgroup per
// network replica, all T steps, workgroup barriers where the stepwise form has kernel boundaries.  The workgroup's threads loop
// over the network's items (ghosts, cells, samples, vehicle slots: a few per thread for the grids the reference builds).  No
// cross-workgroup exchange exists, so nothing spins; replicas are independent workgroups (BASELINE config 5's pattern for networks
// beyond the fused kernels' one-item-per-thread limits).  What fits the workgroup's LDS stays there for the episode -- ns_plan picks
// the instantiation <TA, SS, MS>: TA = 3 the static tables, the per-step table rows (double-buffered, fetched one step ahead by the
// last wavefronts), ghosts / slots / ghost cotangents, the step's signal table, the routes; SS = 2 the state rows and cotangent
// planes, 1 the planes alone; MS the micro side's running state -- and is reached through address_space(3) pointers (ds_read); the
// rest through address_space(1) pointers into the workspace and the history (tape rows, event list, running-mean streams).
// A step costs its phases' instruction issue on one compute unit: ~17 + 14 us at 252 lanes / 1 152 cells / 28 IDM lanes (DESIGN
// section 9 lists the phases) against ~80 us of launches in the stepwise form.
// =====================================================================================================================
__device__ __forceinline__ void ns_replica_shift(NsCommon &a, int rep) {       // the arrays replica `rep` owns
    a.hist += (size_t)rep * (a.T + 1) * 4 * (a.C > 0 ? a.C : 1);
    a.queue += (size_t)rep * a.T * a.L;
    if (a.reward) a.reward += (size_t)rep * 2;
    if (a.counts) a.counts += (size_t)rep * 4;
    a.ws += (size_t)rep * a.lo.total;
    const size_t ts = (size_t)rep * (size_t)a.table_stride;
    a.left_src += ts; a.left_gate += ts; a.right_src += ts; a.conv_next += ts; a.schedule += ts;
    if (a.draws) a.draws += (size_t)rep * (size_t)a.draws_stride;
}

// One cell of step t: BOTH its interfaces (ARZ.riemann_solve + the two Jacobian products each, redone by the neighbouring cell's
// thread: a solve costs less than handing its result over), Godunov update, float32 glue, the cell's three blocks (MacroLane.forward,
// _macro_lane.py:83-146; dMacroLane._backward, dmacro_lane.py:96-132) -- the operations of the straight-lane operator's kernel
// kc0: the interface constants that depend on the speed limit and dt alone (the persistent kernels evaluate them once, in LDS)
template <class A> __device__ __forceinline__ void ns_cell_item(const A &a, const IfaceConst &kc0, int t, int c, int &fault_step, int &fault_lane,
                                                                int &fault_index) {
    const int C = a.C;
    const int l = a.cell_lane[c];
    const int off = a.lane_off[l], n = a.lane_ncell[l];
    const int k = c - off;
    const double dx = a.lane_dx[l];
    double cc;
    IfaceConst kc = kc0;
    if constexpr (A::kTB) {
        cc = a.lane_cc[l];
        kc.dx = dx; kc.cfl_lim = a.lane_cfl[l]; kc.cfl_lim2 = 2.0 * kc.cfl_lim; kc.lim_ok = 1e-5 < kc.cfl_lim;
    } else {
        cc = a.dt_d / dx;
        kc.set_grid(a.dt_d, dx);
    }
    const auto cur = ns_state(a, t);
    const auto nxt = ns_state(a, t + 1);
    const auto gh = ns_ghosts(a) + (size_t)a.lane_gpos[l] * 8;
    const double r0 = cur[c], y0 = cur[C + c], u0 = cur[2 * C + c], q0 = cur[3 * C + c];
    double rL, yL, uL, qL, rR, yR, uR, qR;
    if (k == 0) {
        const float g0 = gh[0];
        if (ghost_is_source(g0)) ghost_source_unpack(gh[2], gh[3], a.um_d, rL, yL, uL, qL);       // a source lane: (r, u) in double
        else { rL = g0; yL = gh[1]; uL = gh[2]; qL = gh[3]; }
    }
    else { rL = cur[c - 1]; yL = cur[C + c - 1]; uL = cur[2 * C + c - 1]; qL = cur[3 * C + c - 1]; }
    if (k == n - 1) { rR = gh[4]; yR = gh[5]; uR = gh[6]; qR = gh[7]; }
    else { rR = cur[c + 1]; yR = cur[C + c + 1]; uR = cur[2 * C + c + 1]; qR = cur[3 * C + c + 1]; }
    Iface fl, fr;
    arz_interface(rL, yL, uL, qL, r0, y0, u0, q0, kc, fl);
    arz_interface(r0, y0, u0, q0, rR, yR, uR, qR, kc, fr);
    if (fault_step < 0 && (fl.cfl_bad || (k == n - 1 && fr.cfl_bad))) { fault_step = t; fault_lane = l; fault_index = fl.cfl_bad ? k : n; }
    const float nr = (float)(r0 + (fl.Fr - fr.Fr) * cc);
    const float ny = (float)(y0 + (fl.Fy - fr.Fy) * cc);
    float nu, nq;
    glue_from_r_y(nr, ny, a.um, nu, nq);
    nxt[c] = nr; nxt[C + c] = ny; nxt[2 * C + c] = nu; nxt[3 * C + c] = nq;
    if constexpr (A::kST) {             // (the state rows live in LDS: the history -- what the reverse sweep and the callers read -- is written here)
        const auto hn = ns_glob(a.hist) + (size_t)(t + 1) * 4 * C;
        hn[c] = nr; hn[C + c] = ny; hn[2 * C + c] = nu; hn[3 * C + c] = nq;
    }
    if (!a.hard) {
        const float cf = (float)cc, ncf = (float)(-cc);
        float4 d0, d1, d2;
        d0.x = ncf * (-fl.A[0]); d0.y = ncf * (-fl.A[1]); d0.z = ncf * (-fl.A[2]); d0.w = ncf * (-fl.A[3]);
        d2.x = ncf * fr.B[0]; d2.y = ncf * fr.B[1]; d2.z = ncf * fr.B[2]; d2.w = ncf * fr.B[3];
        d1.x = 1.f - cf * (fr.A[0] - fl.B[0]); d1.y = 0.f - cf * (fr.A[1] - fl.B[1]);
        d1.z = 0.f - cf * (fr.A[2] - fl.B[2]); d1.w = 1.f - cf * (fr.A[3] - fl.B[3]);
        auto tp = ns_ptr<float4>(a, a.lo.ptape) + ((size_t)t * C + c) * 3;
        tp[0] = d0; tp[1] = d1; tp[2] = d2;
    }
}
// reverse of the cells of step t: g[c] <- dqs[c][1]^T g[c] + dqs[c-1][2]^T g[c-1] + dqs[c+1][0]^T g[c+1] inside a lane
// (dMacroForwardLayer.backward, dmacro_lane.py:277-309); the lane's edge cells leave the ghosts' cotangents
template <class A> __device__ __forceinline__ void ns_cell_bwd_item(const A &a, int t, int c) {
    const int C = a.C;
    const int l = a.cell_lane[c];
    const int off = a.lane_off[l], n = a.lane_ncell[l];
    const int k = c - off;
    const auto Gn = ns_G(a, t + 1);
    const auto Gp = ns_G(a, t);
    auto tp = ns_ptr<float4>(a, a.lo.ptape) + ((size_t)t * C + c) * 3;
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
    const auto gg = ns_gghost(a) + (size_t)a.lane_gpos[l] * 4;
    if (k == 0) { gg[0] = (double)dot2(d0.x, gr, d0.z, gy); gg[1] = (double)dot2(d0.y, gr, d0.w, gy); }
    if (k == n - 1) { gg[2] = (double)dot2(d2.x, gr, d2.z, gy); gg[3] = (double)dot2(d2.y, gr, d2.w, gy); }
    if (!(isfinite(pr) && isfinite(py))) net_fault(a.err, DHTS_FAULT_NAN, t, l, k);
}

// The static tables of a network (per lane, per cell, adjacency) staged in LDS for a persistent kernel: every item of every phase
// starts with a chain of dependent look-ups (item -> lane -> offsets -> state), each link a global-memory latency otherwise.
__host__ __device__ inline size_t ns_al16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t ns_stage_bytes(const NsCommon &a) {
    size_t n = 0;
#define NS_X(name, ty, count) n += ns_al16(sizeof(ty) * (size_t)((count) > 0 ? (count) : 1));
    NS_TABLES(NS_X)
#undef NS_X
    return n;
}
__host__ __device__ inline size_t ns_micro_state_bytes(const NsCommon &a, bool bwd) {
    const size_t plane = (size_t)a.Lm * a.cap;
    size_t n = 0;
#define NS_X(name, ty, count) n += ns_al16(sizeof(ty) * (size_t)((count) > 0 ? (count) : 1));
    if (bwd) { NS_MICRO_BWD(NS_X) } else { NS_MICRO_FWD(NS_X) }
#undef NS_X
    return n;
}
// what a persistent kernel carves out of its dynamic LDS behind the phases' scratch (host and device agree through this plan):
// byte sizes; tables + rows + misc are staged together (TB) or not at all, st + gl likewise (ST)
struct NsPlan { int scratch, tables, rows, misc, st, gl, ms, routes; };
template <int TA, int SS, bool MS>
__device__ __forceinline__ void ns_carve(NsArgsT<TA, SS, MS> &a, const NsArgs &a0, NS_LDS char *lds, const NsPlan &pl, bool bwd) {
    NS_LDS char *p = lds + pl.scratch;
    if constexpr (TA == 3) {
        NS_LDS char *q = p;
#define NS_X(name, ty, count) { const size_t n_ = (size_t)((count) > 0 ? (count) : 1); NS_LDS ty *d_ = (NS_LDS ty *)q; \
            if ((count) > 0) for (size_t i_ = threadIdx.x; i_ < n_; i_ += blockDim.x) d_[i_] = a0.name[i_]; \
            a.name = d_; q += ns_al16(sizeof(ty) * n_); }
        NS_TABLES(NS_X)
#undef NS_X
        p += pl.tables;
        a.row_d = (NS_LDS double *)p; a.row_i = (NS_LDS int32_t *)(p + ns_al16(16 * (size_t)a.L));
        if (!bwd) {
            NS_LDS double *cc_ = (NS_LDS double *)(p + ns_al16(16 * (size_t)a.L) + ns_al16(32 * (size_t)a.L)), *cfl_ = cc_ + a.L;
            for (int l = threadIdx.x; l < a.L; l += blockDim.x) { const double dx = a0.lane_dx[l]; cc_[l] = a.dt_d / dx; cfl_[l] = dx / a.dt_d; }
            a.lane_cc = cc_; a.lane_cfl = cfl_;
        } else { a.lane_cc = nullptr; a.lane_cfl = nullptr; }
        p += pl.rows;
        const size_t lg = ns_al16(32 * (size_t)a.L);
        if (bwd) { a.gg = (NS_LDS double *)p; a.sl = (NS_LDS float *)(p + lg); a.gh = nullptr; }
        else { a.gh = (NS_LDS float *)p; a.gg = nullptr; a.sl = nullptr; }
        a.sg = (NS_LDS float *)(p + (bwd ? 2 : 1) * lg); a.sgi = (NS_LDS int32_t *)(a.sg + 8 * (size_t)a.sq);
        a.ow = bwd ? nullptr : (NS_LDS float *)(p + lg + ns_al16(32 * (size_t)a.sq + 16));
        p += pl.misc;
    } else {
#define NS_X(name, ty, count) a.name = (NS_GLOBAL const ty *)a0.name;
        NS_TABLES(NS_X)
#undef NS_X
        a.gh = nullptr; a.sl = nullptr; a.gg = nullptr; a.row_i = nullptr; a.row_d = nullptr; a.sg = nullptr; a.sgi = nullptr; a.ow = nullptr;
    }
    a.st = nullptr; a.gl = nullptr;
    if constexpr (SS == 2) { a.st = (NS_LDS float *)p; p += pl.st; }
    if constexpr (SS >= 1) { a.gl = bwd ? (NS_LDS float *)p : nullptr; p += pl.gl; }
    if constexpr (TA == 3) {
        a.routes_l = nullptr; a.rlen_l = nullptr;
        if (pl.routes) {
            const size_t n = (size_t)a.n_routes * a.route_stride;
            NS_LDS int32_t *r_ = (NS_LDS int32_t *)p, *len_ = (NS_LDS int32_t *)(p + ns_al16(4 * n));
            const auto src = ns_glob(a0.routes);
            for (size_t i = threadIdx.x; i < n; i += blockDim.x) r_[i] = src[i];
            for (int r = threadIdx.x; r < a.n_routes; r += blockDim.x) {
                int k = 0;
                while (k < a.route_stride && src[(size_t)r * a.route_stride + k] >= 0) k++;
                len_[r] = k;
            }
            a.routes_l = r_; a.rlen_l = len_;
            p += pl.routes;
        }
    }
    if constexpr (MS) {
        const size_t plane = (size_t)a.Lm * a.cap;
#define NS_X(name, ty, count) a.m_##name = (NS_LDS ty *)p; p += ns_al16(sizeof(ty) * (size_t)((count) > 0 ? (count) : 1));
        if (bwd) { NS_MICRO_BWD(NS_X) } else { NS_MICRO_FWD(NS_X) }
#undef NS_X
    }
}
// the replica's argument block in LDS (`a_s`): everybody copies the tables, thread 0 writes the scalars and the pointers
template <int TA, int SS, bool MS>
__device__ __forceinline__ void ns_persist_setup(NsArgsT<TA, SS, MS> &a_s, const NsArgs &a0, NS_LDS char *lds, const NsPlan &pl, bool bwd) {
    NsArgsT<TA, SS, MS> tmp;
    static_cast<NsCommon &>(tmp) = static_cast<const NsCommon &>(a0);
    ns_replica_shift(tmp, blockIdx.x);
    ns_carve(tmp, a0, lds, pl, bwd);
    if (threadIdx.x == 0) a_s = tmp;
    __syncthreads();
}
// the table rows of step r into their LDS copy (r & 1): every thread fetches its elements (registers `v`), stores them later
struct NsRowRegs { int32_t i[4]; double d; };
template <class A> __device__ __forceinline__ void ns_rows_fetch(const A &a, int r, int l, NsRowRegs &v) {
    const size_t o = (size_t)r * a.L + l;
    v.i[0] = ns_glob(a.left_src)[o]; v.i[1] = ns_glob(a.left_gate)[o]; v.i[2] = ns_glob(a.right_src)[o]; v.i[3] = ns_glob(a.conv_next)[o];
    v.d = ns_glob(a.schedule)[o];
}
template <class A> __device__ __forceinline__ void ns_rows_store(const A &a, int r, int l, const NsRowRegs &v) {
    const auto b = a.row_i + (size_t)(r & 1) * 4 * a.L;
    b[l] = v.i[0]; b[a.L + l] = v.i[1]; b[2 * a.L + l] = v.i[2]; b[3 * a.L + l] = v.i[3];
    a.row_d[(size_t)(r & 1) * a.L + l] = v.d;
}

template <int TA, int SS, bool MS>
__global__ void __launch_bounds__(kNsBlock) ns_persist_fwd_kernel(NsArgs a0, const float *__restrict__ action_all, NsPlan pl) {
    extern __shared__ double lds_p[];
    // the replica's arguments -- ~100 pointers and offsets, the staged ones as LDS pointers -- live in LDS themselves: as a local copy
    // they cost 487 scalar-register spills and 79 vector ones (scratch traffic inside every phase)
    constexpr bool ST = SS == 2;
    __shared__ NsArgsT<TA, SS, MS> a_s;
    __shared__ IfaceConst kc0_s;
    if (threadIdx.x == 0) { IfaceConst k0; k0.set_um(a0.um_d); k0.set_grid(a0.dt_d, 1.0); kc0_s = k0; }
    ns_persist_setup(a_s, a0, (NS_LDS char *)reinterpret_cast<char *>(lds_p), pl, false);
    const NsArgsT<TA, SS, MS> &a = a_s;
    const float *action = action_all + (size_t)blockIdx.x * a0.n_action;
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a0.L, C = a0.C, T = a0.T;
    // the episode's running state: empty road (MacroLane.__init__), stored ghosts (0, u_max), no vehicles, counters zero
    {
        const auto s0 = ns_state(a, 0);
        for (int i = tid; i < C; i += B) {
            s0[i] = 0.f; s0[C + i] = 0.f; s0[2 * C + i] = a.um; s0[3 * C + i] = a.um;
            if constexpr (ST) { const auto h0 = ns_glob(a.hist); h0[i] = 0.f; h0[C + i] = 0.f; h0[2 * C + i] = a.um; h0[3 * C + i] = a.um; }
        }
    }
    for (int i = tid; i < L; i += B) {
        const auto o = ns_ptr<float>(a, a.lo.own_hist); o[2 * i] = 0.f; o[2 * i + 1] = a.um;
        if constexpr (TA == 3) { a.ow[2 * i] = 0.f; a.ow[2 * i + 1] = a.um; }
    }
    if constexpr (MS) {
        NS_LDS unsigned *z = (NS_LDS unsigned *)a.m_P;
        const size_t nz = (size_t)pl.ms / 4;
        for (size_t i = tid; i < nz; i += B) z[i] = 0u;
    } else {
        auto z = ns_ptr<unsigned>(a, a.lo.P);
        const size_t nz = (a.lo.counters + sizeof(NsCounters) - a.lo.P) / 4;
        for (size_t i = tid; i < nz; i += B) z[i] = 0u;
    }
    if constexpr (TA == 3) for (int l = tid; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, 0, l, v); ns_rows_store(a, 0, l, v); }
    for (int q = tid; q < a0.sq; q += B) ns_signal_fill(a, action, 0, q);
    __syncthreads();
    int fault_step = -1, fault_lane = 0, fault_index = 0;
#ifdef DHTS_NS_STAMPS
    if (threadIdx.x < 24) ns_sub_[threadIdx.x] = 0;
    __syncthreads();
    long long st_[6] = {0, 0, 0, 0, 0, 0}, st_last_ = __builtin_amdgcn_s_memtime();
#define NS_STAMP(i) { const long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_last_; st_last_ = n_; }
#else
#define NS_STAMP(i)
#endif
    for (int t = 0; t < T; ++t) {
        // the next step's table rows into the copy this step does not read -- by the MIDDLE wavefronts (lane l by thread B / 2 + l), which
        // have neither a ghost item (the first wavefronts) nor a head gap (the last): held in registers across the step they were
        // spilled, five dependent global loads at the head of every step
        if constexpr (TA == 3) {
            if (t + 1 < T) for (int l = (tid + B / 2) % B; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, t + 1, l, v); ns_rows_store(a, t + 1, l, v); }
        }
        if (t + 1 < T) for (int q = B - 1 - tid; q < a0.sq; q += B) ns_signal_fill(a, action, t + 1, q);      // (the last wavefront: idle in the ghost phase)
        for (int j = tid; j < ns_ghost_items(L); j += B) ns_ghost_fwd_item(a, t, action, j);
        NS_STAMP(0)
        ns_micro_fwd(a, t, action, reinterpret_cast<float *>(lds_p));
        __syncthreads();
        NS_STAMP(1)
        for (int c = tid; c < C; c += B) ns_cell_item(a, kc0_s, t, c, fault_step, fault_lane, fault_index);
        __syncthreads();
        NS_STAMP(2)
        ns_convert(a, t, reinterpret_cast<int *>(lds_p));
        NS_STAMP(3)
        __syncthreads();
        NS_STAMP(4)
    }
#ifdef DHTS_NS_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        printf("ns_persist_fwd<%d,%d> cycles per step: ghosts %lld | micro boundary + IDM %lld | cells %lld | hand-offs + loss %lld | flush %lld\n",
               TA, SS + 4 * (int)MS, st_[0] / T, st_[1] / T, st_[2] / T, st_[3] / T, st_[4] / T);
    if (blockIdx.x == 0 && tid == 0)
        printf("   micro: admission %lld | head gaps %lld | IDM %lld;  hand-offs: capacitors %lld | event walk %lld | loss %lld\n",
               ns_sub_[0] / T, ns_sub_[1] / T, ns_sub_[2] / T, ns_sub_[3] / T, ns_sub_[4] / T, ns_sub_[5] / T);
    if (blockIdx.x == 0 && tid == 0)
        printf("   head gaps: lane part %lld | scans %lld (rest in 'head gaps');  loss: lane prefixes %lld | samples (thread 0) %lld | wait %lld | lane sums = 'loss'\n",
               ns_sub_[16] / T, ns_sub_[17] / T, ns_sub_[6] / T, ns_sub_[7] / T, ns_sub_[15] / T);
#endif
    if (fault_step >= 0) net_fault(a.err, DHTS_FAULT_CFL, fault_step, fault_lane, fault_index);
    if constexpr (MS) {                  // what the reverse sweep starts from
        auto lane_n = ns_ptr<int>(a, a.lo.lane_n);
        for (int m = tid; m < a0.Lm; m += B) lane_n[m] = a.m_lane_n[m];
        if (tid == 0) *ns_ptr<NsCounters>(a, a.lo.counters) = *a.m_counters;
    }
    ns_reward(a, reinterpret_cast<float *>(lds_p));
}

template <int TA, int SS, bool MS>
__global__ void __launch_bounds__(kNsBlock) ns_persist_bwd_kernel(NsArgs a0, const float *__restrict__ action_all, const float *__restrict__ g_reward,
                                                                float *__restrict__ g_action_all, NsPlan pl) {
    extern __shared__ double lds_p[];
    constexpr bool ST = SS == 2, SG = SS >= 1;
    __shared__ NsArgsT<TA, SS, MS> a_s;                             // (see ns_persist_fwd_kernel)
    ns_persist_setup(a_s, a0, (NS_LDS char *)reinterpret_cast<char *>(lds_p), pl, true);
    const NsArgsT<TA, SS, MS> &a = a_s;
    const float *action = action_all + (size_t)blockIdx.x * a0.n_action;
    const int tid = threadIdx.x, B = blockDim.x;
    const int L = a0.L, C = a0.C, T = a0.T, Lm = a0.Lm;
    {   // cotangents start at zero; the lanes hold what the forward left
        auto z = ns_ptr<unsigned>(a, a.lo.G);
        const size_t nz = (a.lo.n_bwd - a.lo.G) / 4;
        for (size_t i = tid; i < nz; i += B) z[i] = 0u;
        if constexpr (SG) for (int i = tid; i < 6 * C; i += B) a.gl[i] = 0.f;
        if constexpr (MS) {
            NS_LDS unsigned *zl = (NS_LDS unsigned *)a.m_gP;
            for (size_t i = tid; i < (size_t)pl.ms / 4; i += B) zl[i] = 0u;
            __syncthreads();
        }
        auto n_bwd = NS_MS(a, int, n_bwd);
        auto lane_n = ns_ptr<int>(a, a.lo.lane_n);
        for (int m = tid; m < Lm; m += B) n_bwd[m] = lane_n[m];
    }
    if constexpr (ST) {                  // rows T and T - 1 of the history
        for (int r = T; r >= T - 1 && r >= 0; --r) {
            const auto sr = ns_state(a, r);
            auto hr = ns_glob(a.hist) + (size_t)r * 4 * C;
            for (int i = tid; i < 4 * C; i += B) sr[i] = hr[i];
        }
    }
    if constexpr (TA == 3) for (int l = tid; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, T - 1, l, v); ns_rows_store(a, T - 1, l, v); }
    for (int q = tid; q < a0.sq; q += B) ns_signal_fill(a, action, T - 1, q);
    __syncthreads();
#ifdef DHTS_NS_STAMPS
    if (threadIdx.x < 24) ns_sub_[threadIdx.x] = 0;
    __syncthreads();
    long long st_[6] = {0, 0, 0, 0, 0, 0}, st_last_ = __builtin_amdgcn_s_memtime();
#endif
    const int n_max = ns_ptr<NsCounters>(a, a.lo.counters)->n_max;
    for (int t = T - 1; t >= 0; --t) {
        if constexpr (TA == 3) {             // (see ns_persist_fwd_kernel)
            if (t >= 1) for (int l = B - 1 - tid; l < L; l += B) { NsRowRegs v; ns_rows_fetch(a, t - 1, l, v); ns_rows_store(a, t - 1, l, v); }
        }
        if (t >= 1) for (int q = B - 1 - tid; q < a0.sq; q += B) ns_signal_fill(a, action, t - 1, q);
        ns_micro_bwd(a, t, action, g_reward ? g_reward + blockIdx.x : nullptr, lds_p, n_max);
        __syncthreads();
        NS_STAMP(0)
        // the state before step t - 1 (history row t - 1): fetched now (row t + 1, whose LDS copy it replaces, is not read after this
        // point), stored at the end of the step
        float sv[8];
        const bool pre_s = ST && t >= 1 && 4 * C <= 8 * B;
        if (pre_s) {
            const auto hr = ns_glob(a.hist) + (size_t)(t - 1) * 4 * C;
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = tid + k * B; sv[k] = i < 4 * C ? hr[i] : 0.f; }
        }
        for (int c = tid; c < C; c += B) ns_cell_bwd_item(a, t, c);
        __syncthreads();
        NS_STAMP(1)
        if (C > 0) {
            for (int j = tid; j < ns_ghost_items(L); j += B) ns_ghost_bwd_item(a, t, action, j);
            __syncthreads();
            NS_STAMP(2)
            for (int m = tid; m < L; m += B) ns_ghost_gather_item(a, t, m);
            for (int q = (B >> 6) - 1 - (tid >> 6); q < a.sq; q += B >> 6) ns_action_gather_wave(a, q);      // (from the last wavefront down)
            NS_STAMP(3)
        }
        // the state before step t - 1 (row t - 1) into the copy row t + 1 leaves; the rows of step t - 1
        if constexpr (ST) {
            if (pre_s) {
                const auto sr = ns_state(a, t - 1);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int i = tid + k * B; if (i < 4 * C) sr[i] = sv[k]; }
            } else if (t >= 1) {
                const auto sr = ns_state(a, t - 1);
                auto hr = ns_glob(a.hist) + (size_t)(t - 1) * 4 * C;
                for (int i = tid; i < 4 * C; i += B) sr[i] = hr[i];
            }
        }
        __syncthreads();
        NS_STAMP(4)
    }
#ifdef DHTS_NS_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        printf("ns_persist_bwd<%d,%d> cycles per step: taps + events + fold + IDM + head gaps %lld | cells %lld | ghosts %lld | gather %lld | state %lld\n",
               TA, SS + 4 * (int)MS, st_[0] / T, st_[1] / T, st_[2] / T, st_[3] / T, st_[4] / T);
    if (blockIdx.x == 0 && tid == 0)
        printf("   taps %lld | events %lld | capacitor charges %lld | fold %lld | IDM %lld | head gaps %lld | admission %lld\n",
               ns_sub_[8] / T, ns_sub_[9] / T, ns_sub_[10] / T, ns_sub_[11] / T, ns_sub_[12] / T, ns_sub_[13] / T, ns_sub_[14] / T);
#endif
    float *g_action = g_action_all + (size_t)blockIdx.x * a.n_action;
    auto g_act = NS_MS(a, double, g_act);
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
    if (i < a.L) { const auto o = ns_ptr<float>(a, a.lo.own_hist); o[2 * i] = 0.f; o[2 * i + 1] = a.um; }
}

}  // namespace dhts

using namespace dhts;

// Hand-off events an episode may record (64 B each; the reverse sweep replays them).  A step holds at most one leaving head per
// micro lane (CHANGE / DESPAWN / DEPOSIT + a DEPCELL per cell the deposit touches) and one SPAWN (+ CAPSERIAL) per capacitor;
// the default budgets a quarter of the lanes doing so in every step -- 8 per step at least, which is what small networks always
// had -- and a caller that meets DHTS_FAULT_CAPACITY with index -2 raises dhts_netstep_tables::max_events (ItscpEnv does, up to
// the hard bound T x (4 n_micro + 2 n_caps)).
static inline int ns_max_events(const dhts_net_desc *d, const dhts_netstep_tables *t) {
    if (t->max_events > 0) return t->max_events;
    const long long per_step = (t->hyb.n_micro + t->n_caps + 3) / 4;
    const long long n = (long long)d->n_steps * (per_step > 8 ? per_step : 8) + 64;
    return n > 0x3fffffff ? 0x3fffffff : (int)n;
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
    return (d->n_replicas == 1 || (t->persistent && d->n_replicas > 1)) && (d->n_cells == 0 || t->cell_lane) &&
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
    a.cell_lane = t->cell_lane; a.table_stride = h.net.replica_stride; a.draws_stride = h.draws_stride;
    a.n_edges = h.net.n_edges; a.n_islots = t->n_inter_slots; a.has_source = h.lane_source != nullptr; a.n_routes = h.n_routes; a.tensor_ladder = h.micro_tensor_ladder; a.veh_params = h.veh_params;
    { int lg = 0; for (int g = 0; g < t->n_groups; ++g) lg += t->groups[g].n_lanes; a.NI = d->n_cells + lg; }
    a.ws = reinterpret_cast<char *>(ws); a.lo = ns_layout(d, t);
    a.hist = hist; a.queue = queue; a.reward = reward; a.counts = counts; a.err = err;
    return a;
}

// What a workgroup's 160 KB hold beside the phases' scratch: the static tables + the per-step table rows + ghosts resp. slots / ghost
// cotangents (TB: small, the head of every item's look-up chain), then the state rows and -- reverse sweep -- the cotangent planes (ST)
int dhts_netstep_block = 0;            // DHTS_OPT_NETSTEP_BLOCK: threads per workgroup of the persistent kernels (0 = heuristic)
int dhts_netstep_lds_kb = 0;           // DHTS_OPT_NETSTEP_LDS_KB (dhts_set_option, macro_kernels.hip): 0 = all a workgroup may take
static NsPlan ns_plan(const NsArgs &a, size_t scratch, bool bwd) {
    NsPlan pl = {(int)scratch, 0, 0, 0, 0, 0, 0, 0};
    const size_t budget = (size_t)(dhts_netstep_lds_kb > 0 ? dhts_netstep_lds_kb : 158) * 1024;      // (the kernels' static LDS -- argument block, scan scratch -- stays below 2 KB)
    size_t left = budget > scratch ? budget - scratch : 0;
    const size_t tables = ns_stage_bytes(a), rows = ns_al16(16 * (size_t)a.L) + ns_al16(32 * (size_t)a.L) + (bwd ? 0 : ns_al16(16 * (size_t)a.L));
    const size_t misc = (bwd ? 2 : 1) * ns_al16(32 * (size_t)a.L) + ns_al16(32 * (size_t)a.sq + 16) + (bwd ? 0 : ns_al16(16 * (size_t)a.L));
    if (tables + rows + misc <= left) { pl.tables = (int)tables; pl.rows = (int)rows; pl.misc = (int)misc; left -= tables + rows + misc; }
    const size_t routes = ns_al16(4 * (size_t)a.n_routes * a.route_stride) + ns_al16(4 * (size_t)a.n_routes);
    if (pl.tables && !bwd && a.Lm > 0 && a.n_routes > 0 && routes <= 32 * 1024 && routes <= left) { pl.routes = (int)routes; left -= routes; }
    const size_t ms = ns_micro_state_bytes(a, bwd);
    if (pl.tables && a.Lm > 0 && ms <= left) { pl.ms = (int)ms; left -= ms; }
    const size_t st = ns_al16(sizeof(float) * 8 * (size_t)a.C), gl = bwd ? ns_al16(sizeof(float) * 6 * (size_t)a.C) : 0;
    if (a.C > 0 && st + gl <= left) { pl.st = (int)st; pl.gl = (int)gl; }
    else if (a.C > 0 && bwd && gl <= left) pl.gl = (int)gl;          // (the planes are read AND written by four phases of a step; the state rows only read)
    return pl;
}

// threads of a persistent workgroup: all 1 024 unless asked otherwise
static int ns_persist_block(const NsArgs &a) {
    if (dhts_netstep_block > 0) return dhts_netstep_block;
    return kNsBlock;
}

template <int TA, int SS, bool MS>
static int ns_launch_persist_fwd(const NsArgs &a, const float *action, const NsPlan &pl, size_t lds, int n_replicas, hipStream_t st) {
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void *)ns_persist_fwd_kernel<TA, SS, MS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_INVALID;
    ns_persist_fwd_kernel<TA, SS, MS><<<n_replicas, ns_persist_block(a), lds, st>>>(a, action, pl);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}
template <int TA, int SS, bool MS>
static int ns_launch_persist_bwd(const NsArgs &a, const float *action, const float *g_reward, float *g_action, const NsPlan &pl, size_t lds,
                                 int n_replicas, hipStream_t st) {
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void *)ns_persist_bwd_kernel<TA, SS, MS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_INVALID;
    ns_persist_bwd_kernel<TA, SS, MS><<<n_replicas, ns_persist_block(a), lds, st>>>(a, action, g_reward, g_action, pl);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

// dhts_common.hip: the reward as the reference's one float32 chain, lanes outermost (DHTS_OPT_REWARD_CHAIN)
extern int dhts_opt_reward_chain;
int dhts_launch_reward_chain(int R, int T, int L, const float *queue, const int32_t *lane_macro, int hard, double dt, int loss_steps,
                             float *reward, int stride, void *stream);

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
        if (ns_convert_scratch(L, C, plane) > lds) lds = ns_convert_scratch(L, C, plane);
        const NsPlan pl = ns_plan(a, ns_al16(lds), false);
        lds = (size_t)pl.scratch + pl.tables + pl.st + pl.gl + pl.rows + pl.misc + pl.ms + pl.routes;
        if (lds > 160 * 1024) return DHTS_E_INVALID;
        int rc;
        if (pl.tables && pl.ms) rc = pl.st ? ns_launch_persist_fwd<3, 2, true>(a, action, pl, lds, d->n_replicas, st) : ns_launch_persist_fwd<3, 0, true>(a, action, pl, lds, d->n_replicas, st);
        else if (pl.tables) rc = pl.st ? ns_launch_persist_fwd<3, 2, false>(a, action, pl, lds, d->n_replicas, st) : ns_launch_persist_fwd<3, 0, false>(a, action, pl, lds, d->n_replicas, st);
        else rc = pl.st ? ns_launch_persist_fwd<1, 2, false>(a, action, pl, lds, d->n_replicas, st) : ns_launch_persist_fwd<1, 0, false>(a, action, pl, lds, d->n_replicas, st);
        if (rc == DHTS_OK && dhts_opt_reward_chain)
            rc = dhts_launch_reward_chain(d->n_replicas, T, L, queue, t->hyb.lane_macro, hard, d->dt, a.loss_steps, reward, 2, stream);
        return rc;
    }
    // running state of the episode
    if (hipMemsetAsync(ws + a.lo.P, 0, a.lo.counters + sizeof(NsCounters) - a.lo.P, st) != hipSuccess) return DHTS_E_LAUNCH;
    const int nmax = C > L ? C : L;
    ns_init_fwd_kernel<<<(nmax + 255) / 256, 256, 0, st>>>(a);
    const size_t lds_b = sizeof(float) * 2 * (size_t)(Lm > 0 ? Lm : 1), lds_c = ns_convert_scratch(L, C, plane);
    if (lds_b > 48 * 1024 || lds_c > 160 * 1024 ||
        (lds_c > 48 * 1024 && hipFuncSetAttribute((const void *)ns_convert_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess))
        return DHTS_E_INVALID;
    const int ghost_blocks = (ns_ghost_items(L) + kNsBlock - 1) / kNsBlock;
    float *ghost = reinterpret_cast<float *>(ws + a.lo.ghost);
    float *tape0 = reinterpret_cast<float *>(ws + a.lo.tape);
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
    if (hipGetLastError() != hipSuccess) return DHTS_E_LAUNCH;
    if (dhts_opt_reward_chain) return dhts_launch_reward_chain(1, T, L, queue, t->hyb.lane_macro, hard, d->dt, a.loss_steps, reward, 2, stream);
    return DHTS_OK;
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
        const NsPlan pl = ns_plan(a, ns_al16(ns_micro_bwd_scratch(L, C, Lm, (size_t)Lm * a.cap)), true);
        const size_t lds = (size_t)pl.scratch + pl.tables + pl.st + pl.gl + pl.rows + pl.misc + pl.ms + pl.routes;
        if (lds > 160 * 1024) return DHTS_E_INVALID;
#define NS_BWD(TA_, SS_, MS_) ns_launch_persist_bwd<TA_, SS_, MS_>(a, action, g_reward, g_action, pl, lds, d->n_replicas, st)
        const int ss = pl.st ? 2 : (pl.gl ? 1 : 0);
        if (pl.tables && pl.ms) return ss == 2 ? NS_BWD(3, 2, true) : (ss == 1 ? NS_BWD(3, 1, true) : NS_BWD(3, 0, true));
        if (pl.tables) return ss == 2 ? NS_BWD(3, 2, false) : (ss == 1 ? NS_BWD(3, 1, false) : NS_BWD(3, 0, false));
        return ss == 2 ? NS_BWD(1, 2, false) : (ss == 1 ? NS_BWD(1, 1, false) : NS_BWD(1, 0, false));
#undef NS_BWD
    }
    // cotangents start at zero; the lanes hold what the forward left (lane_n)
    if (hipMemsetAsync(ws + a.lo.G, 0, a.lo.n_bwd - a.lo.G, st) != hipSuccess) return DHTS_E_LAUNCH;
    if (Lm > 0 && hipMemcpyAsync(ws + a.lo.n_bwd, ws + a.lo.lane_n, sizeof(int) * (size_t)Lm, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return DHTS_E_LAUNCH;
    const size_t lds_m = ns_micro_bwd_scratch(L, C, Lm, (size_t)Lm * a.cap);
    if (lds_m > 160 * 1024 ||
        hipFuncSetAttribute((const void *)ns_micro_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m) != hipSuccess)
        return DHTS_E_INVALID;
    float *G = reinterpret_cast<float *>(ws + a.lo.G);
    double *g_ghost = reinterpret_cast<double *>(ws + a.lo.g_ghost);
    const float *tape0 = reinterpret_cast<const float *>(ws + a.lo.tape);
    const int lane_blocks = (L + 255) / 256, act_blocks = (a.sq + 3) / 4;
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
            ns_ghosts_bwd_kernel<<<(ns_ghost_items(L) + 255) / 256, 256, 0, st>>>(a, step, action);
            ns_ghosts_gather_kernel<<<lane_blocks + act_blocks, 256, 0, st>>>(a, step, lane_blocks);
        }
    }
    ns_finish_bwd_kernel<<<(a.n_action + 255) / 256, 256, 0, st>>>(a, g_action);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

}  // extern "C"
