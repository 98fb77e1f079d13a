// hybrid_kernels.hip -- time-fused forward and reverse sweeps of a HYBRID road network (itscp `hybrid` mode): ARZ cell
// lanes, IDM vehicle lanes and the hand-offs between them, one workgroup per network replica, on gfx950.
//
// Reference per step (example/control/itscp/_env.py:620-768; _simulator.py:56-276; road/network/road_network.py:79-170,
// 429-580; road/network/conversion.py:11-215): signals -> boundaries of every lane (macro: blended ghost cells, micro: head
// gap of the head vehicle) -> one step per lane -> commit -> hand-offs in lane-id order -> queue-length loss.
//
// Work split inside the workgroup: the macro side runs one item per thread exactly like network_kernels.hip (ghosts,
// interface solves, cell updates, ordered prefix mean).  The micro side is a handful of vehicles on a few lanes: an extra
// wavefront (the "micro wave") takes it, lane j of that wave owning micro lane j (head gap, IDM steps, loss terms, commits)
// and flux capacitor j; only the hand-off events themselves (spawn, lane change, despawn, deposit: rare, order-dependent)
// are walked serially by lane 0.  All micro state lives in LDS.  Every float32 operation of the micro side is written,
// with its partial derivatives, to a per-replica record stream in HBM: each lane stages its records in LDS and the wave
// flushes them once per step (lane-ordered, coalesced; a lane's block = its segments head gaps + IDM steps | capacitors +
// events | loss + commits), noting the per-lane counts of every segment in an index.  The reverse kernel loads a
// step's records back into LDS and lets every lane replay its own segments backwards (the same thing torch autograd does
// for the reference) in step with the hand-written macro adjoint; the two sides meet at the hand-off records (capacitor
// reads, deposits) and at the signals.  Contributions that cross lanes (the leader of a head vehicle sits on another lane;
// several lanes look at the same signal) go through per-lane outboxes that lane 0 applies in lane order, so the result does
// not depend on timing (no atomics).
//
// Ids on the record stream: [0, 3V) = persistent slots of the vehicles (position, speed, ancillary a), [3V, 3V + 16) =
// flux capacitors, above that the step's temporaries: a private range per lane of the micro wave (+ one for the events),
// reused every step; COMMIT records copy temporaries into the slots at the end of a step and every temporary's adjoint is
// cleared when its defining record is replayed, so the reverse kernel keeps one step's adjoints in LDS.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "../../include/dhts.h"
#include "arz_device.hpp"
#include "idm_device.hpp"
#include "net_device.hpp"

namespace dhts {

// first thread of the downstream-ghost group: the next wavefront boundary behind the L upstream-ghost threads when both groups
// fit in front of the micro wavefront (the block's last 64 threads), else right behind them (tid < 2 L)
__host__ __device__ inline int hyb_ghost_base1(int L, int B) {
    const int p = (L + 63) & ~63;
    return (p + L <= B - 64) ? p : L;
}
constexpr int kMaxMicro = 64;        // micro lanes per network: lane j of the micro wave owns micro lane j
constexpr int kMaxCaps = 16;         // macro lanes with a micro successor
// (vehicles per micro lane: 16 unless the tables ask for more -- dhts_hybrid_tables::lane_capacity = 32, 64, 128 -- a per-launch LDS size)
constexpr int kMaxVeh = 128;         // vehicles per replica and episode
constexpr int kRouteStride = 32;     // MAX_ROUTE_LENGTH, road_network.py:17
constexpr int kLaneLocals = 192;     // temporaries per lane of the micro wave and step
constexpr int kEventLocals = 64;     // temporaries of the serial event walk per step
// (a launch lays out loc_lanes * kLaneLocals + kEventLocals temporaries: HybTables::loc_lanes = 64, or the staging lanes when packed)
constexpr int kStageH = 48;          // records a lane can stage per step, at most.  A lane's staging area is two blocks of stage_h
                                     // records (the block being filled, the block being flushed); stage_h is a per-launch size:
                                     // kStageH when the LDS has room for that (<= 20 micro lanes beside BASELINE config 4's 256
                                     // cells), less for networks with more micro lanes (hyb_stage_h)
constexpr int kPhases = 3;           // record segments per block (= step): head gaps + IDM | capacitors + events |
                                     // commits + the loss seeds of the state the step leaves
constexpr int kMaxStepRecords = 1024;
constexpr int kWindow = 100000;      // RunningMean(100_000), _env.py:122

// Phase timing for tools/probes/exp_hyb_stamps.py: compiled in with -DDHTS_HYB_STAMPS only (a separate build of the library,
// never the product).  Every barrier of a step stamps s_memtime before and behind itself; work(i) = arriving at barrier i -
// leaving the one before, drain(i) = the wave's own outstanding LDS / scalar operations, barrier(i) = the rest; summed over the steps, written by the first lane of every wavefront
// of the first eight replicas.
#ifdef DHTS_HYB_STAMPS
__device__ long long dhts_hyb_stamps[2][8][16][24];
// replica 0, every step: (arrival, release) of every wavefront at every barrier, cycles since the kernel's first stamp
__device__ int dhts_hyb_trace[2][640][16][10];
#define HYB_STAMP_DECL long long st_last_ = __builtin_amdgcn_s_memtime(), st_acc_[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    const long long st_t0_ = st_last_; int st_kern_ = 0;
#define HYB_STAMP_KERNEL(k_) st_kern_ = (k_);
#define HYB_BARRIER(i)                                                                      \
    {                                                                                       \
        const long long a_ = __builtin_amdgcn_s_memtime();                                  \
        st_acc_[i] += a_ - st_last_;                           /* work */                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  \
        const long long b_ = __builtin_amdgcn_s_memtime();                                  \
        st_acc_[8 + (i)] += b_ - a_;                           /* drain of the wave's own LDS / scalar operations */ \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                     \
        st_last_ = __builtin_amdgcn_s_memtime();                                            \
        st_acc_[16 + (i)] += st_last_ - b_;                    /* barrier */                \
        if (blockIdx.x == 0 && (tid & 63) == 0 && t < 640) {                                 \
            dhts_hyb_trace[st_kern_][t][tid >> 6][2 * (i)] = (int)(a_ - st_t0_);             \
            dhts_hyb_trace[st_kern_][t][tid >> 6][2 * (i) + 1] = (int)(st_last_ - st_t0_);   \
        }                                                                                   \
    }
// a point inside a phase: the time since the last stamp goes to work slot 4 + i (slots 4..7 are free in the forward kernel) and
// is NOT counted in the phase's own work slot
#define HYB_SUB(i)                                                                          \
    {                                                                                       \
        const long long a_ = __builtin_amdgcn_s_memtime();                                  \
        st_acc_[4 + (i)] += a_ - st_last_;                                                  \
        st_last_ = a_;                                                                      \
    }
// a second set of inner points: slots 12 + i (free in the forward kernel: it has four barriers)
#define HYB_SUB2(i)                                                                         \
    {                                                                                       \
        const long long a_ = __builtin_amdgcn_s_memtime();                                  \
        st_acc_[12 + (i)] += a_ - st_last_;                                                 \
        st_last_ = a_;                                                                      \
    }
#define HYB_STAMP_WRITE(kernel_, rep_, tid_, B_)                                            \
    if ((rep_) < 8) {                                                                       \
        const int role_ = ((tid_) & 63) == 0 ? (tid_) >> 6 : -1;   /* one row per wavefront */ \
        if (role_ >= 0 && role_ < 16) for (int q_ = 0; q_ < 24; ++q_) dhts_hyb_stamps[kernel_][rep_][role_][q_] = st_acc_[q_]; \
    }
#else
#define HYB_STAMP_DECL
#define HYB_STAMP_KERNEL(k_)
#define HYB_BARRIER(i) lds_barrier()
#define HYB_SUB(i) {}              // (a statement: the points stand behind `if constexpr (...)`)
#define HYB_SUB2(i) {}
#define HYB_STAMP_WRITE(kernel_, rep_, tid_, B_)
#endif

enum { K_NODE = 1, K_COMMIT = 2, K_DEPOSIT = 3, K_CELLREAD = 4, K_SIGNAL = 5, K_SEED = 6, K_IMPORT = 7, K_IDM = 8 };
// (until round 3 a ninth kind carried the capacitors' per-step charge; it is applied without a record now: HybWs::capru / capflag)

struct HybTables {
    NetTables net;
    const int32_t *lane_macro; const double *lane_len; const int32_t *conv_next; const int32_t *routes; const int32_t *route_ptr;
    int n_routes, route_stride, loss_steps, n_micro;
    int lane_sh;                             // log2 of the vehicles a micro lane holds (4 .. 7)
    int lds_budget;                          // bytes of LDS the forward kernel sizes its staging area for
    int loc_lanes;                           // lanes of the micro wave that own a range of temporaries (64; n_staging_lanes when packed)
    int max_step_records;                    // records a step may hold (the reverse sweep stages a step's records in LDS)
    const int32_t *lane_source; const double *draws; int n_draws; size_t draws_stride;     // micro source lanes (itscp `micro` mode)
    int tensor_ladder;                       // dhts_hybrid_tables::micro_tensor_ladder
    const double *veh_params;                // dhts_hybrid_tables::veh_params ([n_routes][6] beside the route table) or NULL
    // a plain RoadNetwork with given initial state and final-state taps (dhts_net_hybrid_state_rollout_*, include/dhts.h)
    int plain;
    const float *state0, *ghost0; float *veh_out; int *events;
    const float *g_stateT, *g_veh; float *g_state0;
};

// workspace layout of one replica (bytes, all 16-byte aligned)
struct HybWs {
    size_t own_hist, rec_k, rec_i, rec_w, step_off, seg_cnt, xs, capru, capflag, mode, per_replica;
    int rec_cap, V;
};
__host__ __device__ inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline HybWs hyb_ws(int L, int C, int T, int n_routes, int records_per_step) {
    HybWs w;
    w.V = kMaxVeh; (void)n_routes;
    const int rps = records_per_step > 0 ? records_per_step : 512;
    w.rec_cap = T * rps + 64;
    size_t o = 0;
    w.own_hist = o; o += up16(sizeof(float) * (size_t)T * 2 * L);
    w.rec_k = o; o += up16(sizeof(int) * (size_t)w.rec_cap);
    w.rec_i = o; o += up16(sizeof(int4) * (size_t)w.rec_cap);
    w.rec_w = o; o += up16(sizeof(float4) * (size_t)w.rec_cap);
    w.step_off = o; o += up16(sizeof(int) * (size_t)(T + 2));                               // blocks 0..T (+ end)
    w.seg_cnt = o; o += up16(sizeof(unsigned short) * (size_t)(T + 1) * kPhases * 64);
    w.xs = o; o += up16(sizeof(float) * (size_t)T * (size_t)(C + kMaxVeh));
    w.capru = o; o += up16(sizeof(float2) * (size_t)T * kMaxCaps);      // (u, r) every capacitor's charge of a step read
    w.capflag = o; o += up16(sizeof(unsigned) * (size_t)T);             // bit j: capacitor j charged; bit 16 + j: its previous level was a variable
    w.mode = o; o += 16;                                                // (loc_lanes | max_step_records << 8) the forward sweep ran with
    w.per_replica = o;
    return w;
}

// a float32 value, the variable it depends on (id < 0: a constant) and d value / d variable: adding or multiplying by a
// constant and negating only change the scale, so they need no record
struct Tv { float val; int id; float sc; };
__device__ __forceinline__ Tv tv_c(float v) { Tv x; x.val = v; x.id = -1; x.sc = 0.f; return x; }
__device__ __forceinline__ Tv tv_var(float v, int id) { Tv x; x.val = v; x.id = id; x.sc = 1.f; return x; }

struct Rec {               // one lane's handle on its staging area (LDS) and its private range of temporaries
    int *sk; int *si; float *sw;
    int cnt, next_local, cap;
    bool over;
    bool off;              // an evaluation episode keeps no records (nothing will be replayed)
};
__device__ __forceinline__ void rec_push(Rec &R, int kind, int out, int4 in, float4 w) {
    if (R.off) return;
    if (R.cnt >= R.cap) { R.over = true; return; }
    const int c = R.cnt++;
    R.sk[c] = (kind << 24) | (out & 0xffffff);
    *reinterpret_cast<int4 *>(R.si + 4 * c) = in;
    *reinterpret_cast<float4 *>(R.sw + 4 * c) = w;
}
__device__ __forceinline__ Tv tv_leaf(Rec &R, float v) { return tv_var(v, R.next_local++); }
__device__ __forceinline__ Tv tv_node4(Rec &R, float v, Tv a, float wa, Tv b, float wb, Tv c, float wc, Tv d, float wd) {
    if (a.id < 0 && b.id < 0 && c.id < 0 && d.id < 0) return tv_c(v);
    const Tv x = tv_var(v, R.next_local++);
    rec_push(R, K_NODE, x.id, make_int4(a.id, b.id, c.id, d.id), make_float4(wa * a.sc, wb * b.sc, wc * c.sc, wd * d.sc));
    return x;
}
// a fresh variable without inputs that still owns a record (so that its adjoint is cleared on replay)
__device__ __forceinline__ Tv tv_fresh(Rec &R, float v) {
    const Tv x = tv_var(v, R.next_local++);
    rec_push(R, K_NODE, x.id, make_int4(-1, -1, -1, -1), make_float4(0.f, 0.f, 0.f, 0.f));
    return x;
}
__device__ __forceinline__ Tv tv_node2(Rec &R, float v, Tv a, float wa, Tv b, float wb) {
    return tv_node4(R, v, a, wa, b, wb, tv_c(0.f), 0.f, tv_c(0.f), 0.f);
}
// a variable with unit scale (what the per-vehicle / per-lane id tables hold)
__device__ __forceinline__ Tv tv_unit(Rec &R, Tv a) {
    return (a.id < 0 || a.sc == 1.f) ? a : tv_node2(R, a.val, a, 1.f, tv_c(0.f), 0.f);
}
__device__ __forceinline__ Tv tv_affine(float v, Tv a, float sc) { Tv x; x.val = v; x.id = a.id; x.sc = a.id < 0 ? 0.f : sc; return x; }
__device__ __forceinline__ Tv tv_add(Rec &R, Tv a, Tv b) {
    if (b.id < 0) return tv_affine(a.val + b.val, a, a.sc);
    if (a.id < 0) return tv_affine(a.val + b.val, b, b.sc);
    return tv_node2(R, a.val + b.val, a, 1.f, b, 1.f);
}
__device__ __forceinline__ Tv tv_sub(Rec &R, Tv a, Tv b) {
    if (b.id < 0) return tv_affine(a.val - b.val, a, a.sc);
    if (a.id < 0) return tv_affine(a.val - b.val, b, -b.sc);
    return tv_node2(R, a.val - b.val, a, 1.f, b, -1.f);
}
__device__ __forceinline__ Tv tv_mul(Rec &R, Tv a, Tv b) {
    if (b.id < 0) return tv_affine(a.val * b.val, a, a.sc * b.val);
    if (a.id < 0) return tv_affine(a.val * b.val, b, b.sc * a.val);
    return tv_node2(R, a.val * b.val, a, b.val, b, a.val);
}
__device__ __forceinline__ Tv tv_div(Rec &R, Tv a, Tv b) {
    if (b.id < 0) return tv_affine(a.val / b.val, a, a.sc / b.val);
    return tv_node2(R, a.val / b.val, a, 1.f / b.val, b, -((a.val / b.val) / b.val));
}
__device__ __forceinline__ Tv tv_soft(Rec &R, Tv a, float k) {
    const float z = a.val * k;
    const float zc = fminf(fmaxf(z, -16.f), 16.f);
    const float sgm = 1.f / (1.f + expf(-zc));
    const float grad = (z < -16.f || z > 16.f) ? 0.f : sgm * (1.f - sgm) * k;
    return tv_node2(R, sgm, a, grad, tv_c(0.f), 0.f);
}
__device__ __forceinline__ Tv tv_pos_or_zero(Tv a) { return a.val > 0.f ? a : tv_c(0.f); }

// ---- forward-mode duals for the head gap: a value and its gradient w.r.t. the (at most) seven variables it can depend on
// -- the head vehicle's (p, v), the leader's (p, v) and the signals of the previous / current / next lane of the route.  The
// ~25 float32 operations of ItscpRoadNetwork.setup_micro_boundary (_simulator.py:176-262) then leave two linear records per
// output (head_position_delta, head_speed_delta) instead of one record each.
constexpr int kDu = 7;
// Du<M>: bit i of M says the value MAY depend on variable i; components outside M are exact zeros and are neither stored
// nor computed.  Every component follows the same float32 formula as the reference's operator (sum, product, quotient,
// sigmoid rule), component by component, so a narrower mask changes no result.
template <unsigned M> struct Du { float v; float g[kDu]; };
template <unsigned M> __device__ __forceinline__ constexpr bool du_has(int i) { return (M >> i) & 1u; }
__device__ __forceinline__ Du<0> du_c(float v) { Du<0> x; x.v = v;
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = 0.f;
    return x; }
template <int I> __device__ __forceinline__ Du<(1u << I)> du_var(float v, float seed = 1.f) { Du<(1u << I)> x; x.v = v;
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = 0.f;
    x.g[I] = seed;
    return x; }
// widen: the same number under a larger mask
template <unsigned N, unsigned M> __device__ __forceinline__ Du<N> du_as(Du<M> a) { static_assert((M & ~N) == 0, "mask"); Du<N> x; x.v = a.v;
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = du_has<M>(i) ? a.g[i] : 0.f;
    return x; }
template <unsigned A, unsigned B> __device__ __forceinline__ Du<(A | B)> du_add(Du<A> a, Du<B> b) { Du<(A | B)> x; x.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = (du_has<A>(i) && du_has<B>(i)) ? a.g[i] + b.g[i] : (du_has<A>(i) ? a.g[i] : (du_has<B>(i) ? b.g[i] : 0.f));
    return x; }
template <unsigned A, unsigned B> __device__ __forceinline__ Du<(A | B)> du_sub(Du<A> a, Du<B> b) { Du<(A | B)> x; x.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = (du_has<A>(i) && du_has<B>(i)) ? a.g[i] - b.g[i] : (du_has<A>(i) ? a.g[i] : (du_has<B>(i) ? -b.g[i] : 0.f));
    return x; }
template <unsigned A, unsigned B> __device__ __forceinline__ Du<(A | B)> du_mul(Du<A> a, Du<B> b) { Du<(A | B)> x; x.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = (du_has<A>(i) && du_has<B>(i)) ? a.g[i] * b.v + b.g[i] * a.v : (du_has<A>(i) ? a.g[i] * b.v : (du_has<B>(i) ? b.g[i] * a.v : 0.f));
    return x; }
template <unsigned A, unsigned B> __device__ __forceinline__ Du<(A | B)> du_div(Du<A> a, Du<B> b) { Du<(A | B)> x; x.v = a.v / b.v;
    const float ia = 1.f / b.v, ib = -((a.v / b.v) / b.v);
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = (du_has<A>(i) && du_has<B>(i)) ? a.g[i] * ia + b.g[i] * ib : (du_has<A>(i) ? a.g[i] * ia : (du_has<B>(i) ? b.g[i] * ib : 0.f));
    return x; }
// (kInl: net_device.hpp sig_exp -- the exact exponential inline or out of line)
template <bool kInl = false, unsigned A> __device__ __forceinline__ Du<A> du_soft(Du<A> a, float k, bool exact = false) {          // dmath.operation.sigmoid(value, constant=k)
    const float z = a.v * k;
    const float zc = fminf(fmaxf(z, -16.f), 16.f);
    const float sgm = 1.f / (1.f + sig_exp<kInl>(-zc, exact));
    const float grad = (z < -16.f || z > 16.f) ? 0.f : sgm * (1.f - sgm) * k;
    Du<A> x; x.v = sgm;
#pragma unroll
    for (int i = 0; i < kDu; ++i) x.g[i] = du_has<A>(i) ? a.g[i] * grad : 0.f;
    return x; }
template <unsigned A> __device__ __forceinline__ Du<A> du_pos_or_zero(Du<A> a) { return a.v > 0.f ? a : du_as<A>(du_c(0.f)); }   // x if x > 0 else 0.0
// out = sum_i g[i] * x_(id[i]) as records: one node for the vehicle inputs, one for the signals (+ the first node)
template <unsigned A> __device__ __forceinline__ Tv du_emit(Rec &R, Du<A> a, const int *id) {
    bool lo = false, hi = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) lo = lo || (du_has<A>(i) && id[i] >= 0 && a.g[i] != 0.f);
#pragma unroll
    for (int i = 4; i < kDu; ++i) hi = hi || (du_has<A>(i) && id[i] >= 0 && a.g[i] != 0.f);
    if (!lo && !hi) return tv_c(a.v);
    int first = -1;
    if (lo) {
        first = R.next_local++;
        rec_push(R, K_NODE, first, make_int4(id[0], id[1], id[2], id[3]), make_float4(a.g[0], a.g[1], a.g[2], a.g[3]));
        if (!hi) return tv_var(a.v, first);
    }
    const int out = R.next_local++;
    rec_push(R, K_NODE, out, make_int4(first, id[4], id[5], id[6]), make_float4(1.f, a.g[4], a.g[5], a.g[6]));
    return tv_var(a.v, out);
}

// sample `idx - kWindow` of the loss' running-mean stream (written hundreds of steps earlier by another thread of this
// workgroup: read around the vector L1)
__device__ __forceinline__ float stream_load(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS carve-up shared by both kernels' host wrappers
struct HybLds {
    size_t fq, scanw, incl, vsp, vep, s0, s1, g, contrib, ql, sig, lanelen, vp, vv, va, vxold, hdpv, hdvv, capv, qmicro, tailsp,
        cell_lane, iface_lane, cnext, vidp, vidv, vida, vcur, vrlen, vroute, lane_n, lane_veh, hdpi, hdvi, vcp, capi, mslot,
        capof, mlane, cbefore, convlist, linfo, caplast, rused, caplane, capleaf, stg_k, stg_i, stg_w, cnt_s, lfl, vx, lane_new, vdsg, vrow, total;
};
__host__ __device__ inline HybLds hyb_lds(int L, int C, int sq, int V, int NS, int stage_h, int lane_sh) {
    HybLds o; size_t p = 0; const int NI = C + L;
    auto D = [&](size_t n) { size_t r = p; p += 8 * ((n + 1) & ~(size_t)1); return r; };
    auto F = [&](size_t n) { size_t r = p; p += 4 * ((n + 3) & ~(size_t)3); return r; };
    o.fq = D(2 * (size_t)NI); o.scanw = D(32); o.incl = D(2 * (size_t)C); o.vsp = D(kMaxMicro + 1); o.vep = D(kMaxMicro + 1);
    o.s0 = F(4 * (size_t)C); o.s1 = F(4 * (size_t)C); o.g = F(8 * (size_t)L); o.contrib = F(2 * (size_t)C); o.ql = F(L);
    o.sig = F(2 * (size_t)sq); o.lanelen = F(L); o.vp = F(V); o.vv = F(V); o.va = F(V); o.vxold = F(V);
    o.hdpv = F(kMaxMicro); o.hdvv = F(kMaxMicro); o.capv = F(kMaxCaps); o.qmicro = F(2 * kMaxMicro); o.tailsp = F(kMaxMicro);
    o.cnext = F(L); o.vidp = F(V); o.vidv = F(V); o.vida = F(V); o.vcur = F(V);
    o.vrlen = F(V); o.vroute = F((size_t)V * kRouteStride); o.lane_n = F(kMaxMicro); o.lane_veh = F((size_t)NS << lane_sh);
    o.hdpi = F(kMaxMicro); o.hdvi = F(kMaxMicro); o.vcp = F(2 * (kMaxMicro + 1)); o.capi = F(kMaxCaps); o.mslot = F(L); o.capof = F(L);
    o.mlane = F(kMaxMicro); o.cbefore = F(kMaxMicro + 1); o.convlist = F(L); o.linfo = F(L); o.caplast = F(kMaxCaps); o.rused = F(kMaxMicro); o.caplane = F(kMaxCaps); o.capleaf = F(kMaxCaps);
    o.stg_k = F((size_t)NS * 2 * stage_h); o.stg_i = F((size_t)NS * 2 * stage_h * 4); o.stg_w = F((size_t)NS * 2 * stage_h * 4);
    // the cell -> lane and interface -> lane maps exist during set-up only: they lie on the (then still empty) staging area
    o.cell_lane = o.stg_i; o.iface_lane = o.stg_i + 4 * (((size_t)C + 3) & ~(size_t)3);
    o.cnt_s = F(2 * kPhases * 64); o.lfl = F(L); o.vx = F(V); o.lane_new = F(kMaxMicro); o.vdsg = F(V);
    o.vrow = F(V);
    o.total = p;
    return o;
}

// The hybrid kernels' Jacobian tape: per (replica, step) the two 2x2 products (A, B) of every interface slot -- [R][T][NIp][2]
// float4, NIp = cells + lanes rounded up to 64 (interface i of the forward kernel's numbering: cells + the macro lanes in front)
__host__ __device__ inline int hyb_tape_ifaces(int L, int C) { return (C + L + 63) & ~63; }

// records a lane of the micro wave can stage per step: what the LDS has room for beside everything else (0 = does not fit)
__host__ __device__ inline int hyb_stage_h(int L, int C, int sq, int V, int NS, int n_action, int lane_sh, size_t budget = 160 * 1024) {
    const size_t act = up16(sizeof(float) * (size_t)n_action);
    const size_t fixed = hyb_lds(L, C, sq, V, NS, 0, lane_sh).total + act;
    if (fixed + 64 >= budget) return 0;
    int h = (int)((budget - fixed - 64) / ((size_t)NS * 2 * 36));
    // (a lane stages one IDM record and one loss seed per vehicle and step beside its head gap's ~15: lanes that hold more than 16
    // vehicles get room for that, as far as the LDS goes)
    const int hcap = lane_sh > 4 ? (2 << lane_sh) + 32 : kStageH;
    if (h > hcap) h = hcap;
    while (h > 0 && hyb_lds(L, C, sq, V, NS, h, lane_sh).total + act > budget) --h;
    // (the set-up maps lie on the staging area's index block)
    if ((size_t)NS * 2 * h * 16 < 4 * ((((size_t)C + 3) & ~(size_t)3) + (((size_t)C + L + 3) & ~(size_t)3))) return 0;
    return h;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// Four barriers per step (phases A: ghosts | head gaps, B: interface solves | IDM steps, C: cell updates, D: hand-offs | loss
// constants).  Where a step goes (s_memtime stamps, -DDHTS_HYB_STAMPS build, 256 replicas of BASELINE config 4;
// profiles/archive/r03l_hybrid_phase_stamps.log): every phase is set by its slowest role -- the micro wave in A (head gaps, ~4 400 cycles)
// and B (IDM steps, ~3 100), the lanes' cell waves in D (loss constants + lane sums, ~4 000-4 500) -- and a step by their sum,
// ~14 700 cycles.  Round 3 took out what was not arithmetic on those paths: the cells' tape blocks (now formed by the reverse
// sweep from an interface tape the interface threads write), the capacitors' chains of dependent LDS look-ups (static ones in
// registers, one entrance-space word per micro lane), table fetches inside divergent branches: 4.02 -> 3.68 ms.  Measured and
// dropped: phases B and C merged with a DPP hand-off of the right flux, (a) 63 slots per wavefront + a helper thread: config 4's
// 384 slots then need a seventh wavefront, the one that also evaluates the vehicles' loss terms (3.97 ms); (b) 64 slots per
// wavefront, the one seam cell per wavefront updated by the micro wave at the start of D (3.68 ms: the barrier's time comes
// back as the longer phases of the roles that were already last; profiles/archive/r03n_hybrid_seam_variant_stamps.log).
// What the differentiable forward pays beyond the evaluation kernel (2.18 ms: same dynamics, nothing kept), by ablation builds
// (profiles/archive/r03q_hybrid_forward_ablation.log): the micro side's records 0.47 ms (and 0.80 ms of the reverse sweep's 2.65), the
// loss' ordered running mean with its constants 0.36 ms (0.22 in the reverse sweep), tape / history / vehicles' loss terms the
// rest.  -O2 / -Os builds of this file run at the same speed (68 KB of code either way, profiles/archive/r03p_hybrid_opt_levels.log).
// Late in round 3 the step loop was split by role (below: two copies, 256 -> 193 vector registers, no spills): 3.67 -> 3.44 ms, the
// evaluation kernel 2.08 -> 1.80 ms; the numbers above are of the single loop.
// kHard: an EVALUATION episode (ItscpEnv.step(action, False); Trainer.evaluate, trainer.py:94-142): hard signals float(a > progress)
// (_env.py:928-960), the macro downstream ghost takes float(signal > 0.5) (_simulator.py:128-137), a head vehicle takes the green
// gap when the signal of its own lane is >= 0.5 and the red one otherwise (scores 0 / 1 / 0, no running mean: _simulator.py:208-
// 276), a cell / a vehicle is static when its speed is below static_speed (_env.py:607-617, 709-717).  The steps themselves and the hand-offs are the
// same.  Nothing is kept for a reverse sweep: no records, no tape, no history, no loss constants (hist, tape, kc and the
// workspace are not touched and may be NULL).
// kState: the instantiation behind dhts_net_hybrid_state_rollout_fwd (given initial state / stored ghosts, plain-network
// semantics, vehicle table and event log out); the itscp instantiations carry none of its tests.
template <int kMaxBlock, bool kHard, bool kState = false>
__global__ void __launch_bounds__(kMaxBlock) net_hybrid_fwd_kernel(int R_, int L, int C, int T, int sq, int F, int n_action, double dt, double um_d,
                                      double static_speed, double veh_len, HybTables tb, const float *__restrict__ action,
                                      float *__restrict__ hist, float4 *__restrict__ tape, float *__restrict__ kc,
                                      float *__restrict__ queue, float *__restrict__ reward, int *__restrict__ counts,
                                      char *__restrict__ workspace, int records_per_step, dhts_error *err) {
    extern __shared__ double lds_d[];
    char *lds = reinterpret_cast<char *>(lds_d);
    const int rep = blockIdx.x, B = blockDim.x;
#ifdef DHTS_HYB_PERM
    // experiment: which roles share a SIMD (hardware wave w runs on SIMD w % 4): the roles of two wavefronts swapped
    const int hw_ = threadIdx.x >> 6;
    const int tid = (((hw_ == DHTS_HYB_PERM_A) ? DHTS_HYB_PERM_B : (hw_ == DHTS_HYB_PERM_B ? DHTS_HYB_PERM_A : hw_)) << 6) | (threadIdx.x & 63);
#else
    const int tid = threadIdx.x;
#endif
    const int NI = C + L, NIp = hyb_tape_ifaces(L, C);
    const HybWs ws = hyb_ws(L, C, T, tb.n_routes, records_per_step);
    const int V = ws.V;
    const int NS = tb.n_micro > kMaxCaps ? tb.n_micro : kMaxCaps;      // lanes of the micro wave that stage records
    const int lane_sh = tb.lane_sh;                                   // a micro lane's vehicle list: lane_veh[(k << lane_sh) + i]
    const int lane_cap = 1 << lane_sh;
    const int stage_h = hyb_stage_h(L, C, sq, V, NS, n_action, lane_sh, (size_t)tb.lds_budget);
    const HybLds lo = hyb_lds(L, C, sq, V, NS, stage_h, lane_sh);
    double *Fq = reinterpret_cast<double *>(lds + lo.fq), *scanw = reinterpret_cast<double *>(lds + lo.scanw);
    double *incl = reinterpret_cast<double *>(lds + lo.incl), *vsp = reinterpret_cast<double *>(lds + lo.vsp), *vep = reinterpret_cast<double *>(lds + lo.vep);
#define LF(name) reinterpret_cast<float *>(lds + lo.name)
#define LI(name) reinterpret_cast<int *>(lds + lo.name)
    float *S0 = LF(s0), *S1 = LF(s1), *G = LF(g), *contrib = LF(contrib), *ql = LF(ql), *sig = LF(sig), *lanelen = LF(lanelen);
    float *vp = LF(vp), *vv = LF(vv), *va = LF(va), *vxold = LF(vxold), *hdpv = LF(hdpv), *hdvv = LF(hdvv), *capv = LF(capv), *qmicro = LF(qmicro);
    float *tailsp = LF(tailsp);                    // per micro lane: free space at its entrance (tail position - half a vehicle, or the
                                                   // lane's length when it is empty) as of the last IDM step: what a capacitor's spawn test reads
    int *cell_lane_s = LI(cell_lane), *iface_lane_s = LI(iface_lane), *cnext = LI(cnext), *vidp = LI(vidp), *vidv = LI(vidv), *vida = LI(vida);
    int *vcur = LI(vcur), *vrlen = LI(vrlen), *vroute = LI(vroute), *lane_n = LI(lane_n), *lane_veh = LI(lane_veh), *hdpi = LI(hdpi), *hdvi = LI(hdvi);
    int *vcp = LI(vcp), *capi = LI(capi), *mslot = LI(mslot), *capof = LI(capof), *mlane = LI(mlane), *cbefore = LI(cbefore), *convlist = LI(convlist), *linfo = LI(linfo), *caplast = LI(caplast), *rused = LI(rused), *caplane = LI(caplane), *capleaf = LI(capleaf);
    int *stg_k = LI(stg_k), *stg_i = LI(stg_i); float *stg_w = LF(stg_w);
    float *vdsg = LF(vdsg);                        // per vehicle: the derivative of its loss term's sigmoid (flush wave, between its two loops)
    float *vx = LF(vx);                            // per vehicle: static_speed - speed of the state the last step left (loss sample)
    int *lfl = LI(lfl);                            // per lane: first cell | last cell << 16
    int *vrow = LI(vrow);                          // per vehicle: its route row (= its row of tb.veh_params)
    int *cnt_s = LI(cnt_s);                        // [2 blocks][kPhases][64 lanes] staged record counts, micro wave -> flush wave
    int *lane_new = LI(lane_new);                  // per micro lane: 1 = its tail vehicle was admitted at the boundary of the step
                                                   // that runs now: the loss of the previous state does not count it
    const float um = (float)um_d, s0f = (float)static_speed, vlen = (float)veh_len, dtf = (float)dt;
    // the replica's action vector is read every step by the signal threads: staged in LDS (behind the carve-up)
    float *act = reinterpret_cast<float *>(lds + lo.total);
    for (int i = tid; i < n_action; i += B) act[i] = action[(size_t)rep * n_action + i];
    const size_t toff = (size_t)rep * tb.net.table_stride;
    float *hist_r = kHard ? nullptr : hist + (size_t)rep * (T + 1) * 4 * C;
    float4 *tape_r = kHard ? nullptr : tape + (size_t)rep * T * NIp * 2;
    float *kc_r = kHard ? nullptr : kc + (size_t)rep * T * C;
    float *queue_r = queue + (size_t)rep * T * L;
    char *wsr = kHard ? nullptr : workspace + (size_t)rep * ws.per_replica;
    float *own_w = reinterpret_cast<float *>(wsr + ws.own_hist);
    int *step_off = reinterpret_cast<int *>(wsr + ws.step_off);
    unsigned short *seg_cnt = reinterpret_cast<unsigned short *>(wsr + ws.seg_cnt);
    float *xs = reinterpret_cast<float *>(wsr + ws.xs);
    float2 *capru_w = reinterpret_cast<float2 *>(wsr + ws.capru);
    unsigned *capflag_w = reinterpret_cast<unsigned *>(wsr + ws.capflag);
    const int mw = tid - (B - 64);                  // lane of the micro wave (the extra wavefront), negative elsewhere
    const bool in_mw = mw >= 0, is_mt = (mw == 0);
    const int loss_steps = tb.loss_steps > 0 ? tb.loss_steps : T;

    // ---- setup: maps, static lists
    if (tid < L) {
        const int off = tb.net.lane_off[tid], n = tb.net.lane_ncell[tid];
        for (int i = 0; i < n; ++i) cell_lane_s[off + i] = tid;
        lanelen[tid] = (float)tb.lane_len[tid];
        linfo[tid] = tb.net.sig_kind[tid] | (tb.net.inter[tid] << 2);
        lfl[tid] = off | ((off + n - 1) << 16);
        ql[tid] = 0.f;
    }
    if (tid < C) {
        float r0 = 0.f, y0 = 0.f, u0 = um, q0 = um;           // an empty road, or the caller's initial state
        if (kState && tb.state0) { const float *s0 = tb.state0 + (size_t)rep * 4 * C; r0 = s0[tid]; y0 = s0[C + tid]; u0 = s0[2 * C + tid]; q0 = s0[3 * C + tid]; }
        S0[tid] = r0; S0[C + tid] = y0; S0[2 * C + tid] = u0; S0[3 * C + tid] = q0;
        if (!kHard) { hist_r[tid] = r0; hist_r[C + tid] = y0; hist_r[2 * C + tid] = u0; hist_r[3 * C + tid] = q0; }
    }
    __syncthreads();
    int n_micro = 0, n_caps = 0, n_conv = 0, n_macro = 0;
    if (tid == 0) {
        // interface numbering: lane l (macro) owns interfaces [off + (macro lanes before l), ... + n]
        int mac = 0, nm = 0, nc = 0, nv = 0, cells = 0;
        for (int l = 0; l < L; ++l) {
            mslot[l] = -1; capof[l] = -1;
            if (tb.lane_macro[l]) {
                const int off = tb.net.lane_off[l], n = tb.net.lane_ncell[l];
                for (int k = 0; k <= n; ++k) iface_lane_s[off + mac + k] = l;
                ++mac; cells += n;
                bool spawns = false;
                for (int e = tb.net.nxt_ptr[l]; e < tb.net.nxt_ptr[l + 1]; ++e) spawns |= !tb.lane_macro[tb.net.nxt_idx[e]];
                if (spawns) { if (nc < kMaxCaps) { capof[l] = nc; caplast[nc] = off + n - 1; caplane[nc] = l; } ++nc; convlist[nv++] = l; }
            } else {
                if (nm < kMaxMicro) { mslot[l] = nm; mlane[nm] = l; cbefore[nm] = cells; lane_n[nm] = 0; rused[nm] = 0; lane_new[nm] = 0; }
                ++nm; convlist[nv++] = l;
            }
        }
        cbefore[nm < kMaxMicro ? nm : kMaxMicro] = cells;
        scanw[0] = (double)mac; scanw[1] = (double)nm; scanw[2] = (double)nc; scanw[3] = (double)nv;
    }
    __syncthreads();
    n_macro = (int)scanw[0]; n_micro = (int)scanw[1]; n_caps = (int)scanw[2]; n_conv = (int)scanw[3];
    const int NIm = C + n_macro;                     // interfaces that exist
    if (n_micro > kMaxMicro || n_caps > kMaxCaps || n_micro != tb.n_micro) { if (tid == 0) net_fault(err, DHTS_FAULT_CAPACITY, -1, 0, n_micro); return; }
    __syncthreads();
    (void)NI; (void)n_conv;          // (the hand-off walk visits candidates since round 4; convlist stays a set-up table)
    // ---- per-thread roles
    // ghost threads: the upstream ghosts of all lanes first, the downstream ones from the next wavefront boundary on, so that a
    // wavefront runs ONE of the two (quite different) blends instead of both under divergence
    // (B includes the micro wavefront, whose lanes carry no ghost role: the grouped layout must end in front of it -- the
    // fallback tid < 2 L always does, hyb_block sizes the cell waves for it)
    const int g_base1 = hyb_ghost_base1(L, B);
    const bool is_if = tid < NIm, is_cell = tid < C, is_ghost = tid < L || (tid >= g_base1 && tid < g_base1 + L), is_lane = tid < L;
    int i_lane = 0, i_k = 0, i_n = 0, i_off = 0, i_mb = 0;
    IfaceConst kconst;
    kconst.set_um(um_d); kconst.set_grid(dt, 1.0);
    int c_lane = 0, c_mb = 0, c_macb = 0; double c_cc = 0.; float c_dxv = 0.f;
    if (is_cell) {
        c_lane = cell_lane_s[tid]; c_cc = dt / tb.net.lane_dx[c_lane]; c_dxv = (float)tb.net.lane_dx[c_lane] / vlen;
        for (int l = 0; l < c_lane; ++l) { if (tb.lane_macro[l]) ++c_macb; else ++c_mb; }
    }
    if (is_if) {
        i_lane = iface_lane_s[tid]; i_off = tb.net.lane_off[i_lane]; i_n = tb.net.lane_ncell[i_lane];
        for (int l = 0; l < i_lane; ++l) if (tb.lane_macro[l]) ++i_mb;
        i_k = tid - i_off - i_mb;
        kconst.set_grid(dt, tb.net.lane_dx[i_lane]);
    }
    const int g_side = tid >= g_base1 ? 1 : 0, g_lane = is_ghost ? (g_side ? tid - g_base1 : tid) : 0;
    int g_kind = 0, g_inter = 0; bool g_macro = false;
    float own_r = 0.f, own_u = um;               // the lane's stored downstream ghost (side-1 ghost thread)
    float gl_r = 0.f, gl_u = um;                 // the lane's stored upstream ghost (side-0 ghost thread)
    const bool plain = kState && tb.plain != 0;
    if (is_ghost) { g_kind = tb.net.sig_kind[g_lane]; g_inter = tb.net.inter[g_lane]; g_macro = tb.lane_macro[g_lane] != 0; }
    int g_if0 = 0;                               // side-0 ghost thread of an ARZ lane: the lane's first interface (its thread's flux slots)
    if (is_ghost && g_macro && g_side == 0) {
        g_if0 = tb.net.lane_off[g_lane];
        for (int l = 0; l < g_lane; ++l) if (tb.lane_macro[l]) ++g_if0;
    }
    if (kState && is_ghost && tb.ghost0) {
        const float *g0 = tb.ghost0 + ((size_t)rep * L + g_lane) * 4;
        gl_r = g0[0]; gl_u = g0[1]; own_r = g0[2]; own_u = g0[3];
    }
    int n_ev = 0;                                // hand-off events logged so far (thread 0 of the micro wave)
    int *ev_log = (kState && tb.events) ? tb.events + (size_t)rep * 4 * kMaxVeh : nullptr;
    int l_off = 0, l_n = 0, l_ms = -1; bool l_macro = false;
    if (is_lane) { l_off = tb.net.lane_off[tid]; l_n = tb.net.lane_ncell[tid]; l_ms = mslot[tid]; l_macro = tb.lane_macro[tid] != 0; }
    // signals of step 0 (the staged action vector is complete after the barrier above); signal threads count (phase, frame)
    int sig_ph = 0, sig_fr = 0;                  // of the step whose signals are computed next
    // the intersections' signal threads sit in the wavefront in front of the micro wave (little else to do there), not in wave 0
    const int sg_base = sq <= 64 ? (((B >> 6) - 2) << 6) : 0;
    const bool is_sg = tid >= sg_base && tid < sg_base + sq;
    const int sg_q = tid - sg_base;
    if (is_sg) { float we, ns, a, pr; int ai; phase_signal_at<(kMaxBlock <= 768)>(act, n_action, sq, F, 0, 0, sg_q, we, ns, a, pr, ai, kHard, tb.tensor_ladder != 0); sig[2 * sg_q] = we; sig[2 * sg_q + 1] = ns; }
    if (++sig_fr == F) { sig_fr = 0; ++sig_ph; }

    // ---- micro wave state
    int *grk = reinterpret_cast<int *>(wsr + ws.rec_k);
    int4 *gri = reinterpret_cast<int4 *>(wsr + ws.rec_i);
    float4 *grw = reinterpret_cast<float4 *>(wsr + ws.rec_w);
    const int base_local = 3 * V + 2 * kMaxCaps;     // [3V, +16) capacitors, [3V + 16, +16) the speed leaf of each capacitor's charge
    Rec rec;
    rec.next_local = base_local; rec.over = false; rec.off = kHard; rec.cap = stage_h;
    auto rec_select = [&](int b) {                   // this lane's staging half `b`, empty
        const int sl = (in_mw && mw < NS) ? mw : 0;
        rec.sk = stg_k + (sl * 2 + b) * stage_h; rec.si = stg_i + (sl * 2 + b) * stage_h * 4; rec.sw = stg_w + (sl * 2 + b) * stage_h * 4;
        rec.cnt = 0;
    };
    rec_select(0);
    // The records a step stages are flushed to HBM by ANOTHER wavefront (the one in front of the micro wave, which has little
    // to do beside the ghosts) while the micro wave already stages the next step's records in the other half.
    const bool is_fw = (tid >> 6) == (B >> 6) - 2;
    const int fl = tid & 63;
    int rec_n = 0;                                   // records of this replica in HBM so far (flush wave, uniform)
    bool fl_fault = false;
    int spawned = 0, deposits = 0; bool cap_fault = false;
    IdmParams idm;
    idm.a_max = um_d * 1.0; idm.a_pref = um_d * 0.8; idm.v_target = um_d * 0.9; idm.min_space = veh_len * 0.1; idm.time_pref = 0.1; idm.length = veh_len;
    double sig_sum = 0.; long long sig_cnt = 0;          // signal_rms (never reaches its window in one episode)
    if (is_mt) for (int j = 0; j < kMaxCaps; ++j) { capv[j] = 0.f; capi[j] = -1; }
    __syncthreads();
    // end of a step (micro wave): every lane's three segment counts (head gaps + IDM | capacitors + events | commits; the
    // flush wave appends the loss seeds to the third) go to LDS and the lane turns to its other staging half
    int seg_a = 0, seg_b = 0;                        // this lane's staged records at the end of the first / second segment
    auto publish = [&](int blk) {
        const int b = blk & 1;
        cnt_s[(b * kPhases + 0) * 64 + mw] = mw < NS ? seg_a : 0;
        cnt_s[(b * kPhases + 1) * 64 + mw] = mw < NS ? seg_b - seg_a : 0;
        cnt_s[(b * kPhases + 2) * 64 + mw] = mw < NS ? rec.cnt - seg_b : 0;
        rec_select(b ^ 1);
    };
    // flush wave, one barrier later: block `blk` goes to HBM in lane order (coalesced), each lane's three segments back to
    // back; the per-lane segment counts go to the index
    double run_in = 0., run_out = 0.; long long run_cnt = 0;
    // vehicle counts in front of the micro lanes, one buffer per parity of the loss index (the micro wave writes the next
    // one while the cells still evaluate the constants of the previous loss)
    auto vcp_of = [&](int ls) { return vcp + (ls & 1) * (kMaxMicro + 1); };
    // flush wave: vehicle samples of the loss' running mean of the state the last step left (stream positions follow the
    // cells of the lanes in front, lane id order), exclusive prefix sums per micro lane for the loss constants
    auto vehicle_samples = [&](int ls) {
        const int k = fl;
        const int nw = k < n_micro ? lane_new[k] : 0;  // (a vehicle admitted at this step's boundary sits at the tail: not part of the last state)
        const int c = k < n_micro ? lane_n[k] - nw : 0;
        if (kHard) {                                   // is_static of every vehicle: 1.0 if speed < static_speed else 0.0
            for (int i = 0; i < c; ++i) { const int vi = lane_veh[(k << lane_sh) + nw + i]; vx[vi] = vv[vi] < s0f ? 1.f : 0.f; }
            return;
        }
        const int exc = k <= n_micro ? vcp_of(ls)[k] : 0;
        double ssum = 0., esum = 0.;
        for (int i = 0; i < c; ++i) {
            const int vi = lane_veh[(k << lane_sh) + nw + i];
            const long long idx = run_cnt + cbefore[k] + exc + i;
            const float x = s0f - vv[vi];
            float xo = 0.f;
            if (idx >= kWindow) xo = stream_load(xs + (idx - kWindow));
            xs[idx] = x; vxold[vi] = xo; vx[vi] = x;
            ssum += (double)x; esum += (double)xo;
        }
        const double is_ = wave_scan_add(ssum), ie_ = wave_scan_add(esum);
        if (k <= n_micro) { vsp[k] = is_ - ssum; vep[k] = ie_ - esum; }
        if (k == 63 && n_micro == 64) { vsp[64] = is_; vep[64] = ie_; }       // (no lane 64 to hold the totals)
    };
    auto flush_block = [&](int blk) {
        if (kHard) return;
        const int b = blk & 1;
        const int c0 = cnt_s[(b * kPhases + 0) * 64 + fl], c1 = cnt_s[(b * kPhases + 1) * 64 + fl], c2 = cnt_s[(b * kPhases + 2) * 64 + fl];
        const int c = c0 + c1 + c2;
        const int inc = wave_scan_add(c);
        const int total = wave_last(inc), exc = inc - c;
        if (fl == 0) step_off[blk] = rec_n;
        seg_cnt[((size_t)blk * kPhases + 0) * 64 + fl] = (unsigned short)c0;
        seg_cnt[((size_t)blk * kPhases + 1) * 64 + fl] = (unsigned short)c1;
        seg_cnt[((size_t)blk * kPhases + 2) * 64 + fl] = (unsigned short)c2;
        if (rec_n + total > ws.rec_cap) fl_fault = true;
        else if (total > 0) {
            // output slot o <- (staging lane, index): the owner is found with register traffic only, then every slot of the
            // step moves in one batch of LDS reads and global stores
            const unsigned long long mask0 = __ballot(c > 0);
            for (int o = fl; o < ((total + 63) & ~63); o += 64) {
                int q = -1;
                unsigned long long mask = mask0;
                while (mask) {
                    const int s_ = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
                    mask &= mask - 1;
                    const int cs = __builtin_amdgcn_readlane(c, s_), os = __builtin_amdgcn_readlane(exc, s_);
                    if (o >= os && o < os + cs) q = (s_ * 2 + b) * stage_h + (o - os);
                }
                if (q >= 0) {
                    grk[rec_n + o] = stg_k[q];
                    gri[rec_n + o] = *reinterpret_cast<const int4 *>(stg_i + 4 * q);
                    grw[rec_n + o] = *reinterpret_cast<const float4 *>(stg_w + 4 * q);
                }
            }
            rec_n += total;
        }
        if (total > tb.max_step_records) fl_fault = true;
    };

    // ---- micro SOURCE lanes (itscp `micro` mode, _simulator.py:153-174): a micro lane without an upstream lane admits a waiting
    // vehicle at the boundary of a step when it has room for half a vehicle at its entrance and the host's draw is below the
    // step's inflow.  The draws are data (tb.draws: the stream np.random.random() yields, consumed in call order = lane order
    // among the lanes that have room); every lane of the micro wave decides for its own lane, ranks come from ballots.  The
    // boundary of step t + 1 is evaluated at the end of step t (same state: after the hand-offs); the admitted vehicle
    // (position 0, speed 0: constants) enters at the tail and is flagged in lane_new so that the loss of the state step t
    // left, which the next step's phases still evaluate, does not count it.
    const bool has_src = tb.lane_source != nullptr;
    bool my_src = false; int my_rlo = 0, my_rn = 0;
    double my_draw = 2.0, my_sched = 0.;
    int draw_base = 0;
    const double *draws_r = has_src ? tb.draws + (size_t)rep * tb.draws_stride : nullptr;
    if (has_src && in_mw && mw < n_micro) {
        const int m = mlane[mw];
        my_src = tb.lane_source[m] != 0;
        my_rlo = tb.route_ptr[m]; my_rn = tb.route_ptr[m + 1] - my_rlo;
    }
    auto src_prefetch = [&](int step) {              // this lane's candidate draw and the inflow of `step`
        const int di = draw_base + mw;
        my_draw = draws_r[di < tb.n_draws ? di : (tb.n_draws > 0 ? tb.n_draws - 1 : 0)];
        const int m = mw < n_micro ? mlane[mw] : 0;
        my_sched = tb.net.schedule[toff + (size_t)(step < T ? step : T - 1) * L + m];
    };
    // how many lanes below this one have their bit set in a ballot (mbcnt: no 64-bit lane mask kept in registers)
    auto lanes_below = [](unsigned long long m) {
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    };
    auto source_admit = [&](int step) {              // micro wave, all lanes; `step` = the step whose boundary this is
        const int k = mw;
        float room = 0.f;
        if (my_src) room = lane_n[k] ? vp[lane_veh[(k << lane_sh) + 0]] - 0.5f * vlen : lanelen[mlane[k]];
        const bool want = my_src && room > vlen * 0.5f;
        const unsigned long long bw = __ballot(want);
        const int rank = lanes_below(bw);
        const double draw = __shfl(my_draw, rank);   // lane j holds draws[draw_base + j]
        if (want && draw_base + rank >= tb.n_draws) cap_fault = true;
        const bool admit = want && draw < my_sched && rused[k < n_micro ? k : 0] < my_rn;
        draw_base += __popcll(bw);
        const unsigned long long ba = __ballot(admit);
        if (admit) {
            const int vi = spawned + lanes_below(ba);
            if (vi >= V || lane_n[k] >= lane_cap) cap_fault = true;
            else {
                const size_t row = (size_t)(my_rlo + rused[k]);
                ++rused[k];
                vp[vi] = 0.f; vidp[vi] = 3 * vi;      // fresh slots: their adjoints start at zero and nothing upstream reads them
                vv[vi] = 0.f; vidv[vi] = 3 * vi + 1;
                va[vi] = vlen; vida[vi] = -1;
                vcur[vi] = 0;
                int rl_ = 0;
                for (int q = 0; q < kRouteStride; ++q) {
                    const int lid = q < tb.route_stride ? tb.routes[row * tb.route_stride + q] : -1;
                    vroute[vi * kRouteStride + q] = lid;
                    if (lid >= 0 && rl_ == q) rl_ = q + 1;
                }
                vrlen[vi] = rl_; vrow[vi] = (int)row;
                for (int q = lane_n[k]; q > 0; --q) lane_veh[(k << lane_sh) + q] = lane_veh[(k << lane_sh) + q - 1];
                lane_veh[(k << lane_sh) + 0] = vi;
                ++lane_n[k];
            }
        }
        if (k < n_micro) lane_new[k] = admit ? 1 : 0;
        spawned += __popcll(ba);
        src_prefetch(step + 1);                      // consumed one step later
    };

    // capacitor j (lane j of the micro wave): its macro lane, that lane's last cell and the lane's (at most 4) successors with
    // their micro slots stay in registers; the step's successor comes from the table one step ahead.  The spawn test then reads
    // one level of LDS (the capacitor, the successor's entrance space) instead of a chain of six dependent look-ups.
    int cap_lane = 0, cap_last = 0, cap_nid[4] = {-1, -1, -1, -1}, cap_nms[4] = {-1, -1, -1, -1};
    int p_capnext = -1;
    if (in_mw && mw < n_caps) {
        cap_lane = caplane[mw]; cap_last = caplast[mw];
        int q = 0;
        for (int e = tb.net.nxt_ptr[cap_lane]; e < tb.net.nxt_ptr[cap_lane + 1] && q < 4; ++e, ++q) {
            const int b = tb.net.nxt_idx[e];
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i == q) { cap_nid[i] = b; cap_nms[i] = mslot[b]; }
        }
    }
    auto cap_fetch = [&](int t) {                    // conv_next of step t for this capacitor's lane (every thread loads: see fetch)
        p_capnext = tb.conv_next[toff + (size_t)(t < T ? t : T - 1) * L + cap_lane];
    };
    auto cap_target = [&](int m) {                   // micro slot of successor m, or -1 (not a successor / a macro lane)
        int ms = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (m >= 0 && cap_nid[i] == m) ms = cap_nms[i];
        return ms;
    };
    if (T > 0) cap_fetch(0);

    float lane_total = 0.f;
    int fault_step = -1, fault_index = 0;
    // per-step table entries, fetched one step ahead
    // Unconditional loads with clamped indices: a load inside a divergent branch merges with the register's old value, and the
    // copy the register allocator places behind it waits for the load at once -- 2 500 cycles of memory latency per step in
    // front of the ghosts (s_memtime stamps, round 3) instead of a load that has a whole step to arrive.
    int p_src = 0, p_gate = 0, p_cnext = -1; double p_sched = 0.;
    const int f_glane = (is_ghost && g_macro) ? g_lane : 0, f_lane = is_lane ? tid : 0;
    const int32_t *f_srcp = g_side == 0 ? tb.net.left_src : tb.net.right_src;
    auto fetch = [&](int t) {
        const int tt = t < T ? t : (T > 0 ? T - 1 : 0);
        const size_t o = toff + (size_t)tt * L + f_glane;
        p_src = f_srcp[o]; p_gate = tb.net.left_gate[o]; p_sched = tb.net.schedule[o];
        p_cnext = tb.conv_next[toff + (size_t)tt * L + f_lane];
    };
    if (T > 0) fetch(0);

    // ---- the loss of the state produced by step t-1 is evaluated inside the phases of step t (its prefix scan beside the
    //      ghosts, its constants and the vehicles' terms beside the interface solves, its lane sums beside the cell updates);
    //      the records of step t-1 are flushed beside the ghosts of step t
    float x_new = 0.f, x_left = 0.f;
    long long x_idx = 0;
    // the sample that leaves the running mean's window when this cell's next one enters: requested at the top of the phase
    // (every thread, clamped index: see fetch), used behind the ghosts
    auto loss_prefetch = [&](int ls) {
        if (kHard) return;
        x_idx = run_cnt + (is_cell ? tid + vcp_of(ls)[c_mb] : 0);
        x_left = stream_load(xs + (x_idx >= kWindow ? x_idx - kWindow : 0));
    };
    auto loss_scan = [&](const float *st, int ls) {
        if (kHard) return;                            // (no running mean in an evaluation episode)
        if ((tid & ~63) < C) {                        // waves without cells (the micro wave) have nothing to add
            double a = 0., b = 0.;
            if (is_cell) {
                x_new = s0f - st[2 * C + tid];
                const long long idx = x_idx;
                a = (double)x_new;
                if (idx >= kWindow) b = (double)x_left;
                xs[idx] = x_new;
            }
            const double ia = wave_scan_add(a), ib = wave_scan_add(b);
            if (is_cell) { incl[tid] = ia; incl[C + tid] = ib; }
            if ((tid & 63) == 63) { scanw[tid >> 6] = ia; scanw[16 + (tid >> 6)] = ib; }
        }
    };
    double tot_a = 0., tot_b = 0.;
    auto loss_consts = [&](const float *st, int ls) {
        float *contrib_w = contrib + (ls & 1) * C;      // (two buffers: the lane sums of loss ls are taken one step later)
        if (kHard) {
            if (is_cell) contrib_w[tid] = (st[2 * C + tid] < s0f ? 1.f : 0.f) * (st[tid] * c_dxv);
            return;
        }
        const int *vcp_l = vcp_of(ls);
        const int nwc = (C + 63) >> 6;
        tot_a = 0.; tot_b = 0.;
        // the wave totals added in order: the sum in front of this wavefront (a scalar, so is the loop) and the sum of all
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        double base_a = 0., base_b = 0.;
        for (int w = 0; w < nwc; ++w) {
            if (w == wv) { base_a = tot_a; base_b = tot_b; }
            tot_a += scanw[w]; tot_b += scanw[16 + w];
        }
        if (is_cell) {
                const long long n = run_cnt + tid + vcp_l[c_mb] + 1;
                const double pin = run_in + base_a + incl[tid] + vsp[c_mb];
                const double pout = run_out + base_b + incl[C + tid] + vep[c_mb];
                const bool full = n > kWindow;             // one IEEE division, operands selected first
                const double mean = (full ? pin - pout : pin) / (full ? (double)kWindow : (double)n);
                const float kk = 16.f / fabsf((float)mean);
                kc_r[(size_t)ls * C + tid] = kk;
                contrib_w[tid] = soft_switch(x_new, kk) * (st[tid] * c_dxv);
                float *hn = hist_r + (size_t)(ls + 1) * 4 * C;
                hn[tid] = st[tid]; hn[C + tid] = st[C + tid]; hn[2 * C + tid] = st[2 * C + tid]; hn[3 * C + tid] = st[3 * C + tid];
            }
    };
    // flush wave (lane fl = micro lane fl): the vehicles' terms of the same loss, from the samples it took beside the ghosts
    // (the micro wave is moving the vehicles meanwhile).  The seeds belong to the speeds the last step left, i.e. behind the
    // last record of block `blk` = ls: they are appended to that block's third segment, which is flushed one phase later.
    auto micro_loss = [&](int ls, int blk) {
        if (is_fw) {
            const int k = fl;
            if (k < n_micro) {
                float *qm = qmicro + (ls & 1) * kMaxMicro;      // (two buffers: the lane sums of loss ls are taken a step later,
                qm[k] = 0.f;                                    //  beside the vehicles' terms of loss ls + 1)
                const int nw = lane_new[k];
                const int nv = lane_n[k] - nw;
                const int *lv = lane_veh + (k << lane_sh) + nw;
                if (kHard) {
                    float q = 0.f;
                    for (int i = 0; i < nv; ++i) q = q + vx[lv[i]];
                    qm[k] = (float)(((double)q * (double)q) * dt);       // (Python floats there: (n ** 2.0) * dt in double, _env.py:709-738)
                } else if (nv > 0) {
                    const int cb = cbefore[k];
                    double pa = run_in + vsp[k], pb = run_out + vep[k];
                    if (cb > 0) {
                        const int c = cb - 1;
                        for (int w = 0; w < (c >> 6); ++w) { pa += scanw[w]; pb += scanw[16 + w]; }
                        pa += incl[c]; pb += incl[C + c];
                    }
                    long long n = run_cnt + cb + vcp_of(ls)[k];
                    // q = sum_i sigmoid(k_i (s0 - v_i)); term = q^2 dt; d reward / d v_i = -2 dt q * (-sigmoid'_i): one SEED
                    // record per vehicle, directly on its speed
                    float q = 0.f;
                    for (int i = 0; i < nv; ++i) {
                        const int vi = lv[i];
                        const float x = vx[vi];
                        pa += (double)x; pb += (double)vxold[vi]; ++n;
                        const bool full = n > kWindow;
                        const double mean = (full ? pa - pb : pa) / (full ? (double)kWindow : (double)n);
                        const float kk = 16.f / fabsf((float)mean);
                        const float z = x * kk;
                        const float zc = fminf(fmaxf(z, -16.f), 16.f);
                        const float sgm = 1.f / (1.f + expf(-zc));
                        vdsg[vi] = (z < -16.f || z > 16.f) ? 0.f : sgm * (1.f - sgm) * kk;
                        q = q + sgm;
                    }
                    qm[k] = (q * q) * dtf;
                    if (ls < loss_steps) {
                        const float gq = -1.0f * dtf * 2.f * q;
                        const int b = blk & 1;
                        int at = cnt_s[(b * kPhases + 0) * 64 + k] + cnt_s[(b * kPhases + 1) * 64 + k] + cnt_s[(b * kPhases + 2) * 64 + k];
                        int added = 0;
                        for (int i = 0; i < nv; ++i) {
                            const int vi = lv[i];
                            const float dsg_i = vdsg[vi];
                            if (vidv[vi] >= 0 && dsg_i != 0.f) {
                                if (at >= stage_h) { fl_fault = true; break; }
                                const int q_ = (k * 2 + b) * stage_h + at;
                                stg_k[q_] = K_SEED << 24;
                                *reinterpret_cast<int4 *>(stg_i + 4 * q_) = make_int4(vidv[vi], 0, 0, 0);
                                *reinterpret_cast<float4 *>(stg_w + 4 * q_) = make_float4(gq * (-dsg_i), 0.f, 0.f, 0.f);
                                ++at; ++added;
                            }
                        }
                        cnt_s[(b * kPhases + 2) * 64 + k] += added;
                    }
                }
            }
        }
    };
    auto run_update = [&](int ls) { if (kHard) return; run_in += tot_a + vsp[n_micro]; run_out += tot_b + vep[n_micro]; run_cnt += C + vcp_of(ls)[n_micro]; };
    auto loss_lanes = [&](int ls) {
        if (is_lane) {
            float term;
            if (l_macro) {
                // q = the lane's cells added in order (float32, as the reference does); the loads go out eight at a time
                float q = 0.f;
                const float *contrib_r = contrib + (ls & 1) * C;
                for (int i0 = 0; i0 < l_n; i0 += 8) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = contrib_r[l_off + (i0 + j < l_n ? i0 + j : l_n - 1)];
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (i0 + j < l_n) q = q + v[j];
                }
                term = (q * q) * dtf;
            } else term = l_ms >= 0 ? qmicro[(ls & 1) * kMaxMicro + l_ms] : 0.f;
            queue_r[(size_t)ls * L + tid] = term;
            lane_total = lane_total + (-1.0f) * term;
        }
    };
    if (has_src && in_mw && T > 0) { src_prefetch(0); source_admit(0); }
    if (has_src) __syncthreads();
    HYB_STAMP_DECL
    // The step, in two copies: the micro wave's and everybody else's.  Which roles a thread can have is decided by its wavefront,
    // but the compiler cannot know that: in one loop every thread holds the loop-carried state of every role (prefetch registers,
    // running sums, record counters, ...) and the kernel sat at the register limits (256 vector registers, ~380 uniform values
    // for ~100 scalar ones).  The barriers pair up by count, not by place.  (The step's text is included twice rather than
    // wrapped in a generic lambda: the closure of a lambda that calls the lambdas above is not broken up into registers.)
    if (in_mw) {
        // the step is this wavefront's instruction stream (it is the last to arrive in three of the four phases): it goes first
        // wherever it shares its SIMD's issue slots (forward 3.44 -> 3.25 ms at config 4)
#ifndef DHTS_HYB_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        for (int t = 0; t < T; ++t) {
            constexpr bool kMw = true;
#include "hybrid_fwd_step.inc"
        }
    } else {
        for (int t = 0; t < T; ++t) {
            constexpr bool kMw = false;
#include "hybrid_fwd_step.inc"
        }
    }
    HYB_STAMP_WRITE(0, rep, tid, B)
    // loss of the final state, last records
    if (T > 0) {
        const float *fin = (T & 1) ? S1 : S0;
        if (T > 1) loss_lanes(T - 2);
        loss_prefetch(T - 1);
        loss_scan(fin, T - 1);
        if (is_fw) vehicle_samples(T - 1);
        __syncthreads();
        loss_consts(fin, T - 1);
        micro_loss(T - 1, T - 1);
        if (in_mw) { seg_a = 0; seg_b = 0; publish(T); }      // block T stays empty (the reverse sweep still looks at it)
        run_update(T - 1);
        __syncthreads();
        if (is_fw) { flush_block(T - 1); flush_block(T); }
        loss_lanes(T - 1);
    }
    if (is_fw) {
        if (fl == 0) {
            if (!kHard) { step_off[T + 1] = rec_n; *reinterpret_cast<int *>(wsr + ws.mode) = tb.loc_lanes | (tb.max_step_records << 8); }
            counts[4 * rep + 2] = rec_n;
        }
        if (fl_fault) net_fault(err, DHTS_FAULT_CAPACITY, 0, 0, rec_n);
    }
    if (in_mw) {
        if (is_mt) { counts[4 * rep + 0] = spawned; counts[4 * rep + 1] = deposits; counts[4 * rep + 3] = n_ev; }
        if (rec.over || cap_fault) net_fault(err, DHTS_FAULT_CAPACITY, 0, 0, 0);
        if (kState && tb.veh_out) {
            // every vehicle ever spawned: gone (-1) unless a lane still lists it (one wavefront: its stores to a row stay in order)
            float *vo = tb.veh_out + (size_t)rep * 4 * V;
            const int n_sp = __builtin_amdgcn_readfirstlane(spawned);
            for (int vi = mw; vi < n_sp; vi += 64) { vo[4 * vi] = -1.f; vo[4 * vi + 1] = vp[vi]; vo[4 * vi + 2] = vv[vi]; vo[4 * vi + 3] = va[vi]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (mw < n_micro) for (int i = 0; i < lane_n[mw]; ++i) vo[4 * lane_veh[(mw << lane_sh) + i]] = (float)mlane[mw];
        }
    }
    if (is_lane) ql[tid] = lane_total;
    __syncthreads();
    if (tid == 0) {
        float rew = 0.f;
        for (int l = 0; l < L; ++l) rew = rew + ql[l];
        reward[rep] = rew;
    }
    if (fault_step >= 0) net_fault(err, DHTS_FAULT_CFL, fault_step, i_lane, fault_index);
#undef LF
#undef LI
}

// ---------------------------------------------------------------------------------------------------------------------
// reverse
// ---------------------------------------------------------------------------------------------------------------------
struct HybLdsB {
    size_t red, h0, h1, gl, c0, c2, gq, inl, inf, sg, adj, gam, rk, ri, rw, cell_lane, obi, obf, aval, iptr, iidx, lfl, linfo, total;
};
__host__ __device__ inline HybLdsB hyb_lds_b(int L, int C, int sq, int V, int E, int loc_lanes = 64, int max_rec = kMaxStepRecords) {
    HybLdsB o; size_t p = 0;
    auto D = [&](size_t n) { size_t r = p; p += 8 * n; return r; };
    auto F = [&](size_t n) { size_t r = p; p += 4 * ((n + 3) & ~(size_t)3); return r; };
    o.red = D(2);
    o.h0 = F(3 * (size_t)C); o.h1 = F(3 * (size_t)C); o.gl = F(3 * (size_t)C); o.c0 = F(2 * (size_t)C); o.c2 = F(2 * (size_t)C);
    o.gq = F(L); o.inl = F(3 * (size_t)E); o.inf = F(3 * (size_t)E); o.sg = F(6 * (size_t)sq);
    o.adj = F(3 * (size_t)V + 2 * kMaxCaps + (size_t)loc_lanes * kLaneLocals + kEventLocals); o.gam = F(2 * (size_t)sq);
    o.rk = F(max_rec); o.ri = F(4 * (size_t)max_rec); o.rw = F(4 * (size_t)max_rec);
    o.cell_lane = F(C); o.obi = F(64 * 5); o.obf = F(64 * 5);
    o.aval = F(2 * (size_t)L); o.iptr = F((size_t)sq + 1); o.iidx = F(2 * (size_t)L);
    o.lfl = F(L); o.linfo = F(L);
    o.total = p;
    return o;
}

// Phases per step (five barriers; the step's data and records are fetched one iteration ahead): stage the step's rows and
// records in LDS -> loss taps into the cell cotangents -> micro records of the commit / loss / hand-off / capacitor
// segments, newest first (deposits and capacitor reads exchange cotangents with the cells) -> speed cotangents into (r, y)
// and J^T g per cell -> gather inside the lanes, ghost cotangents to inbox slots, signals | the head-gap / IDM segment of
// the micro records and the outboxes -> edge cells take their inboxes.
template <int kMaxBlock, bool kState = false>
__global__ void __launch_bounds__(kMaxBlock) net_hybrid_bwd_kernel(int R_, int L, int C, int T, int sq, int F, int n_action, double dt, double um_d,
                                      double static_speed, double veh_len, HybTables tb, const float *__restrict__ action,
                                      const float *__restrict__ hist, const float4 *__restrict__ tape,
                                      const float *__restrict__ kc, const float *__restrict__ queue,
                                      const float *__restrict__ g_reward, float *__restrict__ g_action,
                                      const char *__restrict__ workspace, int records_per_step, dhts_error *err) {
    extern __shared__ double lds_d[];
    char *lds = reinterpret_cast<char *>(lds_d);
    const int rep = blockIdx.x, tid = threadIdx.x, B = blockDim.x;
    const int E = tb.net.n_edges > 0 ? tb.net.n_edges : 1;
    const HybWs ws = hyb_ws(L, C, T, tb.n_routes, records_per_step);
    const int V = ws.V;
    const HybLdsB lo = hyb_lds_b(L, C, sq, V, E, tb.loc_lanes, tb.max_step_records);
#define LF(name) reinterpret_cast<float *>(lds + lo.name)
    float *H0 = LF(h0), *H1 = LF(h1), *gL = LF(gl), *c0 = LF(c0), *c2 = LF(c2), *inL = LF(inl), *inF = LF(inf);
    float *sg = LF(sg), *adj = LF(adj), *gam = LF(gam), *rw = LF(rw);
    int *rk = reinterpret_cast<int *>(lds + lo.rk), *ri = reinterpret_cast<int *>(lds + lo.ri);
    int *cell_lane_s = reinterpret_cast<int *>(lds + lo.cell_lane);
    int *obi = reinterpret_cast<int *>(lds + lo.obi); float *obf = LF(obf);
    float *aval = LF(aval);
    int *iptr = reinterpret_cast<int *>(lds + lo.iptr), *iidx = reinterpret_cast<int *>(lds + lo.iidx);
    int *lfl = reinterpret_cast<int *>(lds + lo.lfl), *linfo = reinterpret_cast<int *>(lds + lo.linfo);   // lane tables (first | last << 16), (kind | inter << 2)
    const float um = (float)um_d, s0f = (float)static_speed, vlen = (float)veh_len, dtf = (float)dt;
    float *act = reinterpret_cast<float *>(lds + lo.total);      // the replica's action vector, staged in LDS
    for (int i = tid; i < n_action; i += B) act[i] = action[(size_t)rep * n_action + i];
    const size_t toff = (size_t)rep * tb.net.table_stride;
    const float *hist_r = hist + (size_t)rep * (T + 1) * 4 * C;
    const int NIp = hyb_tape_ifaces(L, C);
    const float4 *tape_r = tape + (size_t)rep * T * NIp * 2;
    const float *kc_r = kc + (size_t)rep * T * C;
    const float *queue_r = queue + (size_t)rep * T * L;
    const char *wsr = workspace + (size_t)rep * ws.per_replica;
    const float *own_r = reinterpret_cast<const float *>(wsr + ws.own_hist);
    const int *step_off = reinterpret_cast<const int *>(wsr + ws.step_off);
    const unsigned short *seg_cnt = reinterpret_cast<const unsigned short *>(wsr + ws.seg_cnt);
    const int *grk = reinterpret_cast<const int *>(wsr + ws.rec_k);
    const int4 *gri = reinterpret_cast<const int4 *>(wsr + ws.rec_i);
    const float4 *grw = reinterpret_cast<const float4 *>(wsr + ws.rec_w);
    const float2 *capru_r = reinterpret_cast<const float2 *>(wsr + ws.capru);
    const unsigned *capflag_r = reinterpret_cast<const unsigned *>(wsr + ws.capflag);
    const float gscale = g_reward ? g_reward[rep] : 1.f;
    const int loss_steps = tb.loss_steps > 0 ? tb.loss_steps : T;
    const int g_base1 = hyb_ghost_base1(L, B);      // ghost threads by side, as in the forward kernel
    const bool is_cell = tid < C, is_ghost = tid < L || (tid >= g_base1 && tid < g_base1 + L), is_lane = tid < L;
    const bool in_mw = tid >= B - 64, is_mt = (tid == B - 64);
    const int mw_lane = tid - (B - 64);
    const int n_adj = 3 * V + 2 * kMaxCaps + tb.loc_lanes * kLaneLocals + kEventLocals;
    // (the forward sweep's temporaries and step records must be laid out as this launch expects: DHTS_OPT_HYB_PACK unchanged in between)
    bool bad_mode = *reinterpret_cast<const int *>(wsr + ws.mode) != (tb.loc_lanes | (tb.max_step_records << 8));

    if (is_lane) {
        const int off = tb.net.lane_off[tid], n = tb.net.lane_ncell[tid];
        for (int i = 0; i < n; ++i) cell_lane_s[off + i] = tid;
        lfl[tid] = off | ((off + n - 1) << 16);
        linfo[tid] = tb.net.sig_kind[tid] | (tb.net.inter[tid] << 2);
    }
    for (int k = tid; k < 3 * E; k += B) { inL[k] = 0.f; inF[k] = 0.f; }
    for (int k = tid; k < 2 * L; k += B) aval[k] = 0.f;
    // action cotangent: a ghost's contribution goes to the intersection of its own lane (the gate of an upstream ghost is an
    // approaching lane of the same intersection), so every intersection sums a fixed list of ghost threads in a fixed order
    if (tid == 0) {
        int n = 0;
        for (int q = 0; q < sq; ++q) {
            iptr[q] = n;
            for (int l = 0; l < L; ++l) if (tb.lane_macro[l] && tb.net.inter[l] == q) { iidx[n++] = 2 * l; iidx[n++] = 2 * l + 1; }
        }
        iptr[sq] = n;
    }
    const bool plain = kState && tb.plain != 0;
    for (int k = tid; k < n_adj; k += B) adj[k] = 0.f;
    for (int k = tid; k < 2 * sq; k += B) gam[k] = 0.f;
    if (is_cell) {
        // cotangent of the final state: the caller's tap, if any (the queue loss adds its own below)
        float t_r = 0.f, t_y = 0.f, t_u = 0.f;
        if (kState && tb.g_stateT) { const float *g = tb.g_stateT + (size_t)rep * 3 * C; t_r = g[tid]; t_y = g[C + tid]; t_u = g[2 * C + tid]; }
        gL[tid] = t_r; gL[C + tid] = t_y; gL[2 * C + tid] = t_u;
    }
    __syncthreads();
    // ... and of the vehicles' final (position, speed): their committed slots 3 k, 3 k + 1
    if (kState && tb.g_veh) for (int k = tid; k < V; k += B) { adj[3 * k] = tb.g_veh[((size_t)rep * V + k) * 2]; adj[3 * k + 1] = tb.g_veh[((size_t)rep * V + k) * 2 + 1]; }
    int c_lane = 0, c_first = 0, c_last = 0, c_macb = 0; float c_dxv = 0.f, c_cf = 0.f, c_ncf = 0.f;
    if (is_cell) {
        c_lane = cell_lane_s[tid]; c_first = tb.net.lane_off[c_lane]; c_last = c_first + tb.net.lane_ncell[c_lane] - 1;
        c_dxv = (float)tb.net.lane_dx[c_lane] / vlen;
        const double cc = dt / tb.net.lane_dx[c_lane];        // update_coefficient, _macro_lane.py:99
        c_cf = (float)cc; c_ncf = (float)(-cc);
        for (int l = 0; l < c_lane; ++l) if (tb.lane_macro[l]) ++c_macb;      // the cell's left interface is slot cell + (macro lanes in front)
    }
    const int f_iL = is_cell ? tid + c_macb : 0;
    const int g_side = tid >= g_base1 ? 1 : 0, g_lane = is_ghost ? (g_side ? tid - g_base1 : tid) : 0;
    int g_kind = 0, g_inter = 0, g_off = 0, g_n = 0; bool g_macro = false;
    if (is_ghost) {
        g_kind = tb.net.sig_kind[g_lane]; g_inter = tb.net.inter[g_lane]; g_off = tb.net.lane_off[g_lane]; g_n = tb.net.lane_ncell[g_lane];
        g_macro = tb.lane_macro[g_lane] != 0;
    }
    // static inbox routing (see network_kernels.hip)
    constexpr int kMaxCand = 4, kMaxEnt = 8;
    int cand_src[kMaxCand], cand_pos[kMaxCand], n_cand = 0;
#pragma unroll
    for (int i = 0; i < kMaxCand; ++i) { cand_src[i] = -1; cand_pos[i] = 0; }
    if (is_ghost && g_macro) {
        const int32_t *my_ptr = g_side == 0 ? tb.net.prv_ptr : tb.net.nxt_ptr, *my_idx = g_side == 0 ? tb.net.prv_idx : tb.net.nxt_idx;
        const int32_t *o_ptr = g_side == 0 ? tb.net.nxt_ptr : tb.net.prv_ptr, *o_idx = g_side == 0 ? tb.net.nxt_idx : tb.net.prv_idx;
        for (int e = my_ptr[g_lane]; e < my_ptr[g_lane + 1] && n_cand < kMaxCand; ++e) {
            const int s_lane = my_idx[e];
            int k = o_ptr[s_lane];
            while (o_idx[k] != g_lane) ++k;
#pragma unroll
            for (int i = 0; i < kMaxCand; ++i) if (i == n_cand) { cand_src[i] = s_lane; cand_pos[i] = k; }
            ++n_cand;
        }
    }
    int ent[kMaxEnt], n_ent = 0;
#pragma unroll
    for (int i = 0; i < kMaxEnt; ++i) ent[i] = -1;
    if (is_cell && (tid == c_first || tid == c_last)) {
        int a = tb.net.nxt_ptr[c_lane], a_end = (tid == c_last) ? tb.net.nxt_ptr[c_lane + 1] : a;
        int b = tb.net.prv_ptr[c_lane], b_end = (tid == c_first) ? tb.net.prv_ptr[c_lane + 1] : b;
        while ((a < a_end || b < b_end) && n_ent < kMaxEnt) {
            const int sa = a < a_end ? 2 * tb.net.nxt_idx[a] : 0x7fffffff;
            const int sb = b < b_end ? 2 * tb.net.prv_idx[b] + 1 : 0x7fffffff;
            const int code = sa < sb ? 2 * a : 2 * b + 1;
            if (sa < sb) ++a; else ++b;
#pragma unroll
            for (int i = 0; i < kMaxEnt; ++i) if (i == n_ent) ent[i] = code;
            ++n_ent;
        }
    }
    // final state's history row
    if (is_cell) {
        const float *h = hist_r + (size_t)T * 4 * C;
        float *Hn = (T & 1) ? H1 : H0;
        Hn[tid] = h[tid]; Hn[C + tid] = h[C + tid]; Hn[2 * C + tid] = h[2 * C + tid];
    }
    float gown_r = 0.f, gown_u = 0.f;    // cotangent of the stored downstream ghost (side-1 ghost thread)
    double ga = 0.; int cur_phase = -1;
    // action cotangent of intersection q: a row of 16 lanes sums the intersection's ghost list (stride 16), its lane 15 owns
    // the running sum of the phase; workgroups too small for 16 lanes per intersection keep one thread per intersection
    const bool row_mode = 16 * sq <= B - 64;
    // the rows sit in the wavefronts right in front of the micro wave (the least loaded ones), 64-aligned so that a row of
    // the reduction is a DPP row
    const int row_base = row_mode ? (B - 64) - ((16 * sq + 63) & ~63) : 0;
    const int own_q = row_mode ? ((tid - row_base) >> 4) : tid;
    const bool in_rows = row_mode ? (tid >= row_base && tid < row_base + 16 * sq) : (tid < sq);
    // signal threads: in the wavefront in front of the micro wave, like in the forward kernel
    const int sg_base = sq <= 64 ? (((B >> 6) - 2) << 6) : 0;
    const bool is_sg = tid >= sg_base && tid < sg_base + sq;
    const int sg_q = tid - sg_base;
    const bool is_own = in_rows && (!row_mode || ((tid - row_base) & 15) == 15);
    int rev_ph = T > 0 ? (T - 1) / F : 0, rev_fr = T > 0 ? (T - 1) % F : 0;      // (t / F, t % F) of the step being reversed
    bool over = false, bad_key = false;
    int bad_step = -1;          // first non-finite cotangent this thread meets (reverse order: the latest step)
    if (tid < sq) for (int k = tid; k < n_action; k += sq) g_action[(size_t)rep * n_action + k] = 0.f;
    __syncthreads();

    // ---- data of step t, fetched one iteration ahead
    constexpr int kPre = 3;              // records per lane of the micro wave held in registers (192 per step; more are loaded late)
    float p_hr = 0.f, p_hy = 0.f, p_hu = 0.f, p_kc = 0.f, p_q = 0.f, p_own_r = 0.f, p_own_u = 0.f;
    float4 p_aL = make_float4(0, 0, 0, 0), p_bL = p_aL, p_aR = p_aL, p_bR = p_aL;
    int p_src = 0, p_gate = 0, p_rlo = 0, p_nrec = 0, p_seg[kPhases], p_rk[kPre];
    int4 p_ri[kPre]; float4 p_rw[kPre];
    unsigned p_capflag = 0u; float2 p_capru = make_float2(0.f, 0.f);
    // capacitor j (lane j of the micro wave): the last cell of its macro lane (numbered as in the forward kernel: macro lanes
    // with a micro successor, in lane order)
    int cap_last_r = 0;
    if (in_mw && mw_lane < kMaxCaps) {
        int nc = 0;
        for (int l = 0; l < L; ++l) {
            if (!tb.lane_macro[l]) continue;
            bool spawns = false;
            for (int e = tb.net.nxt_ptr[l]; e < tb.net.nxt_ptr[l + 1]; ++e) spawns |= !tb.lane_macro[tb.net.nxt_idx[e]];
            if (spawns) { if (nc == mw_lane) cap_last_r = tb.net.lane_off[l] + tb.net.lane_ncell[l] - 1; ++nc; }
        }
    }
#pragma unroll
    for (int ph = 0; ph < kPhases; ++ph) p_seg[ph] = 0;
#pragma unroll
    for (int j = 0; j < kPre; ++j) { p_rk[j] = 0; p_ri[j] = make_int4(0, 0, 0, 0); p_rw[j] = make_float4(0, 0, 0, 0); }
    // Every load below is unconditional with a clamped index: a load inside `if (is_cell)` merges with the register's old
    // value at the join, the register allocator then copies parts of the loaded tuple right behind the load, and the
    // s_waitcnt in front of that copy stalls the wave for the full memory latency in every iteration.
    const int f_cell = is_cell ? tid : 0, f_lane = is_lane ? tid : 0, f_glane = (is_ghost && g_macro) ? g_lane : 0;
    const int32_t *f_srcp = g_side == 0 ? tb.net.left_src : tb.net.right_src;
    int n_rlo = 0, n_nrec = 0;           // record range of the step fetched NEXT (its offsets are loaded one call earlier)
    auto fetch_offsets = [&](int t) {    // step_off[t], step_off[t + 1] for the call after this one
        const int tt = t < 0 ? 0 : t;
        const int a = step_off[tt], b = step_off[tt + 1];
        n_rlo = a; n_nrec = b - a;
    };
    auto fetch_cells = [&](int t) {      // what the cells', ghosts' and lanes' threads need of step t
        const int tt = t < 0 ? 0 : t;
        {
            const float *h = hist_r + (size_t)tt * 4 * C;
            p_hr = h[f_cell]; p_hy = h[C + f_cell]; p_hu = h[2 * C + f_cell];
            p_kc = kc_r[(size_t)tt * C + f_cell];
            const float4 *tp = tape_r + ((size_t)tt * NIp + f_iL) * 2;      // interfaces iL and iL + 1 of the cell: 64 contiguous bytes
            p_aL = tp[0]; p_bL = tp[1]; p_aR = tp[2]; p_bR = tp[3];
        }
        p_q = queue_r[(size_t)tt * L + (is_cell ? c_lane : f_lane)];      // cells: the queue term of their own lane
        {
            const size_t o = toff + (size_t)tt * L + f_glane;
            p_src = f_srcp[o]; p_gate = tb.net.left_gate[o];
            p_own_r = own_r[(size_t)tt * 2 * L + 2 * f_glane]; p_own_u = own_r[(size_t)tt * 2 * L + 2 * f_glane + 1];
        }
    };
    auto fetch_micro = [&](int t) {      // the micro wave's: the step's records, segment counts, capacitor pairs
        const int tt = t < 0 ? 0 : t;
        p_rlo = n_rlo; p_nrec = n_nrec;          // offsets of step t arrived during the previous iteration
#pragma unroll
        for (int ph = 0; ph < kPhases; ++ph) p_seg[ph] = seg_cnt[((size_t)tt * kPhases + ph) * 64 + mw_lane];
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const int k = mw_lane + 64 * j;
            const int kk = p_rlo + (k < p_nrec ? k : 0);
            p_rk[j] = grk[kk]; p_ri[j] = gri[kk]; p_rw[j] = grw[kk];
        }
        fetch_offsets(t - 1);
        p_capflag = capflag_r[tt];
        p_capru = capru_r[(size_t)tt * kMaxCaps + (mw_lane < kMaxCaps ? mw_lane : 0)];
    };
    auto fetch = [&](int t) { if (in_mw) fetch_micro(t); else fetch_cells(t); };      // (wave-uniform)
    // block T of the record stream is empty in streams this build writes (the loss seeds of the final state sit at the end
    // of block T - 1); seeds found there are still applied
    if (T > 0 && in_mw) {
        const int b_lo = step_off[T], b_n = step_off[T + 1] - b_lo;
        int c = 0;
#pragma unroll
        for (int ph = 0; ph < kPhases; ++ph) c += seg_cnt[((size_t)T * kPhases + ph) * 64 + mw_lane];
        const int inc = wave_scan_add(c);
        const int lo_ = inc - c;
        if (lo_ + c <= b_n)
            for (int k = lo_; k < lo_ + c; ++k)
                if ((grk[b_lo + k] >> 24) == K_SEED && (!kState || gscale != 0.f)) { const int4 a = gri[b_lo + k]; const float4 b = grw[b_lo + k]; adj[a.x] += gscale * b.x; }
    }
    __syncthreads();
    if (in_mw) fetch_offsets(T - 1);
    fetch(T - 1);
    HYB_STAMP_DECL
    HYB_STAMP_KERNEL(1)
    // (two copies of the step, by role: see the forward kernel)
    if (in_mw) {
        for (int t = T - 1; t >= 0; --t) {
            constexpr bool kMw = true;
#include "hybrid_bwd_step.inc"
        }
    } else {
        for (int t = T - 1; t >= 0; --t) {
            constexpr bool kMw = false;
#include "hybrid_bwd_step.inc"
        }
    }
    HYB_STAMP_WRITE(1, rep, tid, B)
    // the cotangent of the initial state (the loop's last barrier is behind the cells' last write of gL)
    if (kState && tb.g_state0 && is_cell) {
        float *g0 = tb.g_state0 + (size_t)rep * 3 * C;
        g0[tid] = gL[tid]; g0[C + tid] = gL[C + tid]; g0[2 * C + tid] = gL[2 * C + tid];
    }
    if (is_own && T > 0) ga += (double)gam[own_q];       // step 0's outboxes (buffer 0; the loop's last barrier is behind them)
    if (is_own && cur_phase >= 0) g_action[(size_t)rep * n_action + cur_phase * sq + own_q] = (float)ga;
    if (bad_step >= 0) net_fault(err, DHTS_FAULT_NAN, bad_step, rep, tid);
    if ((over && is_mt) || bad_key || (bad_mode && tid == 0)) net_fault(err, DHTS_FAULT_CAPACITY, 0, 0, bad_mode ? -3 : (bad_key ? -2 : 0));
#undef LF
}

}  // namespace dhts

using namespace dhts;

static inline bool hyb_desc_ok(const dhts_net_desc *d) {
    return d && d->n_replicas > 0 && d->n_lanes > 0 && d->n_cells >= 0 && d->n_steps >= 0 && d->n_inter_sq > 0 &&
           d->frames_per_phase > 0 && d->n_action >= d->n_inter_sq && d->n_action <= 960 && d->dt > 0 && d->u_max > 0 &&
           d->vehicle_length > 0 && d->n_cells + d->n_lanes <= 960;
}
static inline bool hyb_tables_ok(const dhts_hybrid_tables *t) {
    const dhts_net_tables *n = t ? &t->net : nullptr;
    return t && n->lane_ncell && n->lane_off && n->sig_kind && n->inter && n->lane_dx && n->left_src && n->left_gate &&
           n->right_src && n->schedule && n->replica_stride >= 0 && n->nxt_ptr && n->nxt_idx && n->prv_ptr && n->prv_idx &&
           n->n_edges >= 0 && t->lane_macro && t->lane_len && t->conv_next && t->routes && t->route_ptr && t->n_routes > 0 &&
           t->route_stride > 0 && t->route_stride <= kRouteStride && t->n_micro >= 0 && t->n_micro <= kMaxMicro &&
           (t->lane_capacity == 0 || t->lane_capacity == 16 || t->lane_capacity == 32 || t->lane_capacity == 64 || t->lane_capacity == 128) &&
           (t->lane_source == nullptr || (t->draws != nullptr && t->n_draws > 0 && t->draws_stride >= 0));
}
static inline int hyb_lane_sh(const dhts_hybrid_tables *t) {       // log2(lane_capacity): 16 (default), 32, 64 or 128 vehicles per micro lane
    return t->lane_capacity >= 128 ? 7 : (t->lane_capacity >= 64 ? 6 : (t->lane_capacity >= 32 ? 5 : 4));
}
static inline HybTables hyb_tables(const dhts_hybrid_tables *t) {
    HybTables h;
    const dhts_net_tables *n = &t->net;
    h.net.lane_ncell = n->lane_ncell; h.net.lane_off = n->lane_off; h.net.sig_kind = n->sig_kind; h.net.inter = n->inter;
    h.net.lane_dx = n->lane_dx; h.net.left_src = n->left_src; h.net.left_gate = n->left_gate; h.net.right_src = n->right_src;
    h.net.schedule = n->schedule; h.net.table_stride = (size_t)n->replica_stride;
    h.net.nxt_ptr = n->nxt_ptr; h.net.nxt_idx = n->nxt_idx; h.net.prv_ptr = n->prv_ptr; h.net.prv_idx = n->prv_idx; h.net.n_edges = n->n_edges;
    h.lane_macro = t->lane_macro; h.lane_len = t->lane_len; h.conv_next = t->conv_next; h.routes = t->routes; h.route_ptr = t->route_ptr;
    h.n_routes = t->n_routes; h.route_stride = t->route_stride; h.loss_steps = t->loss_steps; h.n_micro = t->n_micro; h.lane_sh = hyb_lane_sh(t);
    h.lane_source = t->lane_source; h.draws = t->draws; h.n_draws = t->n_draws; h.draws_stride = (size_t)t->draws_stride;
    h.tensor_ladder = t->micro_tensor_ladder; h.veh_params = t->veh_params;
    h.lds_budget = 160 * 1024; h.loc_lanes = 64; h.max_step_records = kMaxStepRecords;
    h.plain = 0; h.state0 = nullptr; h.ghost0 = nullptr; h.veh_out = nullptr; h.events = nullptr;
    h.g_stateT = nullptr; h.g_veh = nullptr; h.g_state0 = nullptr;
    return h;
}
int dhts_hyb_pack = 2;                           // DHTS_OPT_HYB_PACK (dhts_set_option, macro_kernels.hip): 0 never, 1 always, 2 = more replicas than CUs
constexpr size_t kHybPackBudget = 79 * 1024;     // two workgroups share a compute unit's 160 KB
static inline int hyb_block(const dhts_net_desc *d) {
    int need = d->n_cells + d->n_lanes;
    if (need < 2 * d->n_lanes) need = 2 * d->n_lanes;
    if (need < d->n_action) need = d->n_action;
    return ((need + 63) & ~63) + 64;               // + the micro wavefront
}

static int hyb_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
// How the fused kernels of one (network, batch) are launched -- the same decision for the forward sweep and its reverse (the
// forward's records name temporaries and fill step blocks whose bounds the reverse sweep's LDS plan must share; a workspace
// word carries the forward's choice and the reverse sweep raises DHTS_FAULT_CAPACITY, index -3, on a mismatch).
// packed: TWO replicas per compute unit -- each workgroup plans with half the LDS (a smaller record staging area in the forward
// kernel; temporaries for the network's own micro lanes and a step's records bounded by what the staging area can hold in the
// reverse kernel) and runs the 128-register instantiation (four wavefronts per SIMD).  Taken when the batch has more replicas
// than the device has compute units (one replica per unit is faster per replica: no register spills), the workgroup has at
// most eight wavefronts, the lanes keep their default capacity of 16 vehicles (a caller climbing the capacity ladder gets
// the full staging area back) and both plans fit.  Measured at BASELINE config 4's network, 512 / 1 024 replicas: forward 6.16 ->
// 4.95 / 12.08 -> 9.74 ms, results bit-identical (tools/probes/exp_hyb_pack.py, profiles/r06_hyb_pack.log).
struct HybPlan { bool packed; size_t fwd_budget; int stage_h, loc_lanes, max_rec; size_t lds_fwd, lds_bwd; int block; };
static HybPlan hyb_plan(const dhts_net_desc *d, const dhts_hybrid_tables *t, bool state_io) {
    HybPlan p;
    const HybWs ws = hyb_ws(d->n_lanes, d->n_cells, d->n_steps, t->n_routes, t->records_per_step);
    const int NS = t->n_micro > kMaxCaps ? t->n_micro : kMaxCaps, lane_sh = hyb_lane_sh(t);
    const int E = t->net.n_edges > 0 ? t->net.n_edges : 1;
    const size_t act = up16(sizeof(float) * (size_t)d->n_action);
    p.block = hyb_block(d);
    p.packed = false;
    const int mode = t->two_per_cu > 0 ? 1 : (t->two_per_cu < 0 ? 0 : dhts_hyb_pack);      // the tables' own word first, then the option
    if (!state_io && p.block <= 512 && lane_sh == 4 && (mode == 1 || (mode == 2 && d->n_replicas > hyb_cu_count()))) {
        const int h = hyb_stage_h(d->n_lanes, d->n_cells, d->n_inter_sq, ws.V, NS, d->n_action, lane_sh, kHybPackBudget);
        const int mr = NS * h < kMaxStepRecords ? NS * h : kMaxStepRecords;
        if (h >= 16 && hyb_lds_b(d->n_lanes, d->n_cells, d->n_inter_sq, ws.V, E, NS, mr).total + act <= kHybPackBudget) p.packed = true;
    }
    p.fwd_budget = p.packed ? kHybPackBudget : (size_t)160 * 1024;
    p.stage_h = hyb_stage_h(d->n_lanes, d->n_cells, d->n_inter_sq, ws.V, NS, d->n_action, lane_sh, p.fwd_budget);
    p.loc_lanes = p.packed ? NS : 64;
    p.max_rec = p.packed ? (NS * p.stage_h < kMaxStepRecords ? NS * p.stage_h : kMaxStepRecords) : kMaxStepRecords;
    p.lds_fwd = hyb_lds(d->n_lanes, d->n_cells, d->n_inter_sq, ws.V, NS, p.stage_h, lane_sh).total + act;
    p.lds_bwd = hyb_lds_b(d->n_lanes, d->n_cells, d->n_inter_sq, ws.V, E, p.loc_lanes, p.max_rec).total + act;
    return p;
}

#ifdef DHTS_HYB_STAMPS
extern "C" int dhts_debug_trace(int *out) {       // [2 kernels][640 steps][16 waves][5 barriers x (arrival, release)] of replica 0
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dhts::dhts_hyb_trace), sizeof(int) * 2 * 640 * 16 * 10) == hipSuccess ? 0 : -1;
}
extern "C" int dhts_debug_stamps(long long *out) {       // [2 kernels][8 replicas][16 waves][24]: work 0..7, drain 8..15, barrier 16..23
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dhts::dhts_hyb_stamps), sizeof(long long) * 2 * 8 * 16 * 24) == hipSuccess ? 0 : -1;
}
#endif

// dhts_common.hip: the reward as the reference's one float32 chain, lanes outermost (DHTS_OPT_REWARD_CHAIN)
extern int dhts_opt_reward_chain;
int dhts_launch_reward_chain(int R, int T, int L, const float *queue, const int32_t *lane_macro, int hard, double dt, int loss_steps,
                             float *reward, int stride, void *stream);

extern "C" {

size_t dhts_net_hybrid_tape_bytes(const dhts_net_desc *d) {
    if (!hyb_desc_ok(d)) return 0;
    return sizeof(float4) * 2 * (size_t)d->n_replicas * d->n_steps * hyb_tape_ifaces(d->n_lanes, d->n_cells);
}
size_t dhts_net_hybrid_workspace_bytes(const dhts_net_desc *d, const dhts_hybrid_tables *t) {
    if (!hyb_desc_ok(d) || !t || t->n_routes <= 0) return 0;
    return hyb_ws(d->n_lanes, d->n_cells, d->n_steps, t->n_routes, t->records_per_step).per_replica * (size_t)d->n_replicas;
}

int dhts_net_hybrid_plan(const dhts_net_desc *d, const dhts_hybrid_tables *t, int32_t plan[8]) {
    if (!hyb_desc_ok(d) || !hyb_tables_ok(t) || !plan) return DHTS_E_INVALID;
    const HybPlan p = hyb_plan(d, t, false);
    plan[0] = p.packed ? 1 : 0; plan[1] = p.block; plan[2] = p.stage_h; plan[3] = (int32_t)p.lds_fwd; plan[4] = (int32_t)p.lds_bwd;
    plan[5] = p.loc_lanes; plan[6] = p.max_rec; plan[7] = hyb_cu_count();
    return DHTS_OK;
}

static int hyb_fwd_launch(const dhts_net_desc *d, const dhts_hybrid_tables *t, const dhts_hybrid_state_io *io, const float *action,
                          float *hist, float *tape, float *kc, float *queue, float *reward, int32_t *counts, void *workspace,
                          dhts_error *err, void *stream) {
    if (!hyb_desc_ok(d) || !hyb_tables_ok(t) || !action || !hist || !tape || !kc || !queue || !reward || !counts || !workspace)
        return DHTS_E_INVALID;
    HybTables ht = hyb_tables(t);
    if (io) { ht.plain = io->plain; ht.state0 = io->state0; ht.ghost0 = io->ghost0; ht.veh_out = io->veh_out; ht.events = io->events; }
    const int B = hyb_block(d);
    if (B > 1024) return DHTS_E_INVALID;
    const HybPlan pl = hyb_plan(d, t, io != nullptr);
    const bool pack = pl.packed;
    ht.lds_budget = (int)pl.fwd_budget; ht.loc_lanes = pl.loc_lanes; ht.max_step_records = pl.max_rec;
    if (pl.stage_h < 8) return DHTS_E_INVALID;       // (a lane with one vehicle stages ~10 records in a step with a hand-off)
    const size_t lds = pl.lds_fwd;
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    // (the block-size bound sets the vector registers a thread may take: 256 / 168 / 128)
    auto kern = io ? (B <= 512 ? net_hybrid_fwd_kernel<512, false, true> : (B <= 768 ? net_hybrid_fwd_kernel<768, false, true> : net_hybrid_fwd_kernel<1024, false, true>))
                   : (pack ? net_hybrid_fwd_kernel<1024, false> : (B <= 512 ? net_hybrid_fwd_kernel<512, false> : (B <= 768 ? net_hybrid_fwd_kernel<768, false> : net_hybrid_fwd_kernel<1024, false>)));
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    kern<<<d->n_replicas, B, lds, (hipStream_t)stream>>>(
        d->n_replicas, d->n_lanes, d->n_cells, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max,
        d->static_speed, d->vehicle_length, ht, action, hist, reinterpret_cast<float4 *>(tape), kc, queue, reward,
        counts, reinterpret_cast<char *>(workspace), t->records_per_step, err);
    if (hipGetLastError() != hipSuccess) return DHTS_E_LAUNCH;
    // (the state form's reward is a by-product nobody reads: plain networks have no loss)
    if (dhts_opt_reward_chain && !io) return dhts_launch_reward_chain(d->n_replicas, d->n_steps, d->n_lanes, queue, t->lane_macro, 0, d->dt, 0, reward, 1, stream);
    return DHTS_OK;
}
int dhts_net_hybrid_rollout_fwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, const float *action, float *hist,
                                float *tape, float *kc, float *queue, float *reward, int32_t *counts, void *workspace,
                                dhts_error *err, void *stream) {
    return hyb_fwd_launch(d, t, nullptr, action, hist, tape, kc, queue, reward, counts, workspace, err, stream);
}
int dhts_net_hybrid_state_rollout_fwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, const dhts_hybrid_state_io *io,
                                      const float *action, float *hist, float *tape, float *kc, float *queue, float *reward,
                                      int32_t *counts, void *workspace, dhts_error *err, void *stream) {
    if (!io) return DHTS_E_INVALID;
    return hyb_fwd_launch(d, t, io, action, hist, tape, kc, queue, reward, counts, workspace, err, stream);
}

int dhts_net_hybrid_rollout_eval(const dhts_net_desc *d, const dhts_hybrid_tables *t, const float *action, float *queue,
                                 float *reward, int32_t *counts, dhts_error *err, void *stream) {
    if (!hyb_desc_ok(d) || !hyb_tables_ok(t) || !action || !queue || !reward || !counts) return DHTS_E_INVALID;
    const int B = hyb_block(d);
    if (B > 1024) return DHTS_E_INVALID;
    const HybWs ws = hyb_ws(d->n_lanes, d->n_cells, d->n_steps, t->n_routes, t->records_per_step);
    const int NS = t->n_micro > kMaxCaps ? t->n_micro : kMaxCaps;
    const int stage_h = hyb_stage_h(d->n_lanes, d->n_cells, d->n_inter_sq, ws.V, NS, d->n_action, hyb_lane_sh(t));
    if (stage_h < 8) return DHTS_E_INVALID;       // (a lane with one vehicle stages ~10 records in a step with a hand-off)
    const size_t lds = hyb_lds(d->n_lanes, d->n_cells, d->n_inter_sq, ws.V, NS, stage_h, hyb_lane_sh(t)).total + up16(sizeof(float) * (size_t)d->n_action);
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    auto kern = B <= 512 ? net_hybrid_fwd_kernel<512, true> : (B <= 768 ? net_hybrid_fwd_kernel<768, true> : net_hybrid_fwd_kernel<1024, true>);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    kern<<<d->n_replicas, B, lds, (hipStream_t)stream>>>(
        d->n_replicas, d->n_lanes, d->n_cells, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max,
        d->static_speed, d->vehicle_length, hyb_tables(t), action, nullptr, nullptr, nullptr, queue, reward, counts, nullptr,
        t->records_per_step, err);
    if (hipGetLastError() != hipSuccess) return DHTS_E_LAUNCH;
    if (dhts_opt_reward_chain) return dhts_launch_reward_chain(d->n_replicas, d->n_steps, d->n_lanes, queue, t->lane_macro, 1, d->dt, 0, reward, 1, stream);
    return DHTS_OK;
}

static int hyb_bwd_launch(const dhts_net_desc *d, const dhts_hybrid_tables *t, int plain, const float *action, const float *hist,
                          const float *tape, const float *kc, const float *queue, const float *g_reward, const float *g_stateT,
                          const float *g_veh, float *g_action, float *g_state0, const void *workspace, dhts_error *err, void *stream) {
    if (!hyb_desc_ok(d) || !hyb_tables_ok(t) || !action || !hist || !tape || !kc || !queue || !g_action || !workspace)
        return DHTS_E_INVALID;
    HybTables ht = hyb_tables(t);
    ht.plain = plain; ht.g_stateT = g_stateT; ht.g_veh = g_veh; ht.g_state0 = g_state0;
    const int B = hyb_block(d);
    if (B > 1024) return DHTS_E_INVALID;
    const bool state_io = plain || g_stateT || g_veh || g_state0;
    const HybPlan pl = hyb_plan(d, t, state_io);
    ht.loc_lanes = pl.loc_lanes; ht.max_step_records = pl.max_rec;
    const size_t lds = pl.lds_bwd;
    if (lds > 160 * 1024) return DHTS_E_INVALID;
    auto kern = state_io ? (B <= 512 ? net_hybrid_bwd_kernel<512, true> : (B <= 768 ? net_hybrid_bwd_kernel<768, true> : net_hybrid_bwd_kernel<1024, true>))
                         : (pl.packed ? net_hybrid_bwd_kernel<1024> : (B <= 512 ? net_hybrid_bwd_kernel<512> : (B <= 768 ? net_hybrid_bwd_kernel<768> : net_hybrid_bwd_kernel<1024>)));
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DHTS_E_LAUNCH;
    kern<<<d->n_replicas, B, lds, (hipStream_t)stream>>>(
        d->n_replicas, d->n_lanes, d->n_cells, d->n_steps, d->n_inter_sq, d->frames_per_phase, d->n_action, d->dt, d->u_max,
        d->static_speed, d->vehicle_length, ht, action, hist, reinterpret_cast<const float4 *>(tape), kc, queue,
        g_reward, g_action, reinterpret_cast<const char *>(workspace), t->records_per_step, err);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}
int dhts_net_hybrid_rollout_bwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, const float *action, const float *hist,
                                const float *tape, const float *kc, const float *queue, const float *g_reward,
                                float *g_action, const void *workspace, dhts_error *err, void *stream) {
    return hyb_bwd_launch(d, t, 0, action, hist, tape, kc, queue, g_reward, nullptr, nullptr, g_action, nullptr, workspace, err, stream);
}
int dhts_net_hybrid_state_rollout_bwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, int32_t plain, const float *action,
                                      const float *hist, const float *tape, const float *kc, const float *queue,
                                      const float *g_reward, const float *g_stateT, const float *g_veh, float *g_action,
                                      float *g_state0, const void *workspace, dhts_error *err, void *stream) {
    return hyb_bwd_launch(d, t, plain, action, hist, tape, kc, queue, g_reward, g_stateT, g_veh, g_action, g_state0, workspace, err, stream);
}

}  // extern "C"
