// netstep_kernels.hip -- the ghost exchange of ONE step of a macro road network of any size, forward and adjoint, on gfx950.
//
// The fused network kernels (network_kernels.hip) keep a replica in one workgroup: cells + lanes <= 1024.  A larger network
// (example/control/itscp/_env.py:221-439 builds any grid) runs step by step: these kernels turn the state before a step
// into every lane's two ghost cells (road_network.py:79-111 with the signal blends of _simulator.py:42-60), the lanes then
// step as ONE batch per (cells, cell length) group through dhts_macro_step_fwd -- the straight-lane operator, whose ghosts
// are inputs -- and the adjoint kernels route the operator's ghost cotangents back to the neighbours' edge cells, the
// stored ghosts of sink lanes and the action.  Same float32 arithmetic, in the same order, as the ghost phases of
// net_macro_fwd_kernel / net_macro_bwd_kernel (net_fwd_step.inc, network_kernels.hip); the host side is dhts/batched.py.
// Thread j = 2 lane + side owns ghost (lane, side).  Nothing here is a hot loop: a step of a 360-lane network is 720 threads.
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"
#include "arz_device.hpp"
#include "net_device.hpp"

namespace dhts {

struct NetStepArgs {
    int L, sq, F, n_action, phase_raw, frame;
    float um; double um_d;
    const int32_t *ncell, *off, *kind, *inter;
    const int32_t *left_src, *left_gate, *right_src;       // rows of this step
    const double *schedule;                                 // row of this step
};

// the signal a ghost looks at: west-east / north-south value of intersection `it` and its derivative w.r.t. the action entry
__device__ __forceinline__ void ghost_signal(const NetStepArgs &a, const float *action, int it, int kd, bool hard,
                                             float &s, float &ds_da, int &a_index) {
    float we, ns, av, pr;
    phase_signal_at(action, a.n_action, a.sq, a.F, a.phase_raw, a.frame, it, we, ns, av, pr, a_index, hard);
    s = kd == 1 ? we : ns;
    const float zs = (av - pr) * kSigK;
    const bool sat = hard || zs < -16.f || zs > 16.f;
    // d sigmoid(k x) / d x = s (1 - s) k with the sigmoid values just computed (0 outside the clamp, like the operator)
    ds_da = sat ? 0.f : (kd == 1 ? we * (1.f - we) * kSigK : -(ns * (1.f - ns) * kSigK));
}

// ghost [L][2][4] = (left, right) x (r, y, u, u_eq); own_in / own_out [L][2] = the lanes' stored downstream ghosts (r, u)
__global__ void net_ghosts_fwd_kernel(NetStepArgs a, int hard, const float *__restrict__ action, const float *__restrict__ r,
                                      const float *__restrict__ u, const float *__restrict__ own_in, float *__restrict__ own_out,
                                      float *__restrict__ ghost) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * a.L) return;
    const int lane = j >> 1, side = j & 1;
    const float um = a.um;
    float fr, fu, fy, fq;
    if (side == 0) {
        const int src = a.left_src[lane], gate = a.left_gate[lane];
        if (src < 0) {                 // source lane: Python floats in the reference (_simulator.py:68-71), read as such by the step operator
            ghost_source_pack(a.schedule[lane], fr, fy, fu, fq);
        } else {
            const int last = a.off[src] + a.ncell[src] - 1;
            const float gr = r[last], gu = u[last];
            float s = 1.f;
            if (gate == -1) s = 0.f;
            else if (gate >= 0) {
                const int kd = a.kind[gate];
                if (kd != 0) { float ds; int ai; ghost_signal(a, action, a.inter[gate], kd, hard != 0, s, ds, ai); }
            }
            fr = gr * s + 0.f * (1.0f - s);
            fu = gu * s + um * (1.0f - s);
            glue_from_r_u(fr, fu, um, fy, fq);
        }
    } else {
        const int src = a.right_src[lane];
        float gr = own_in[2 * lane], gu = own_in[2 * lane + 1];
        if (src >= 0) { const int first = a.off[src]; gr = r[first]; gu = u[first]; }
        const int kd = a.kind[lane];
        float sg = 1.f;
        if (kd != 0) { float ds; int ai; ghost_signal(a, action, a.inter[lane], kd, hard != 0, sg, ds, ai); }
        const float s2 = hard ? (sg > 0.5f ? 1.f : 0.f) : soft_switch(sg - 0.5f, kSigK);
        fr = s2 * gr + (1.0f - s2) * 1.0f;
        fu = s2 * gu + (1.0f - s2) * 0.0f;
        glue_from_r_u(fr, fu, um, fy, fq);
        own_out[2 * lane] = fr; own_out[2 * lane + 1] = fu;
    }
    float *g = ghost + (size_t)j * 4;
    g[0] = fr; g[1] = fy; g[2] = fu; g[3] = fq;
}

// adjoint, part 1 (thread = ghost): the operator's cotangent of the ghost's (r, y) -- plus, for a downstream ghost, the
// cotangent of the stored ghost it also is -- through the glue and the blend.  slot [L][2][4] = (cotangent for the source cell's
// r, for its y, the action partial, the action index as a float); g_own_out [L][2] = cotangent of the stored ghost the lane came in with.
__global__ void net_ghosts_bwd_kernel(NetStepArgs a, const float *__restrict__ action, const float *__restrict__ r,
                                      const float *__restrict__ y, const float *__restrict__ u, const float *__restrict__ own_in,
                                      const double *__restrict__ g_ghost, const float *__restrict__ g_own_in,
                                      float *__restrict__ g_own_out, float *__restrict__ slot) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * a.L) return;
    const int lane = j >> 1, side = j & 1;
    const float um = a.um;
    float add_r = 0.f, add_y = 0.f, a_val = 0.f; int a_key = -1;
    const float gg_r = (float)g_ghost[(size_t)j * 2], gg_y = (float)g_ghost[(size_t)j * 2 + 1];
    if (side == 0) {
        const int src = a.left_src[lane], gate = a.left_gate[lane];
        if (src >= 0) {
            const int last = a.off[src] + a.ncell[src] - 1;
            const float grn_r = r[last], grn_u = u[last];
            float s = 1.f, ds = 0.f; int kd = 0, ai = -1;
            if (gate == -1) s = 0.f;
            else if (gate >= 0) {
                kd = a.kind[gate];
                if (kd != 0) ghost_signal(a, action, a.inter[gate], kd, false, s, ds, ai);
            }
            const float fr = grn_r * s + 0.f * (1.0f - s), fu = grn_u * s + um * (1.0f - s);
            float g_fr = gg_r, g_fu = 0.f;
            glue_y_bwd(fr, fu, um, gg_y, g_fr, g_fu);
            add_r = g_fr * s;
            glue_u_bwd(r[last], y[last], um, g_fu * s, add_r, add_y);
            if (kd != 0) { a_val = (g_fr * grn_r + g_fu * (grn_u - um)) * ds; a_key = ai; }
        }
    } else {
        const int src = a.right_src[lane];
        const int first = src < 0 ? 0 : a.off[src];
        const float grn_r = src < 0 ? own_in[2 * lane] : r[first];
        const float grn_u = src < 0 ? own_in[2 * lane + 1] : u[first];
        const int kd = a.kind[lane];
        float sg = 1.f, ds = 0.f; int ai = -1;
        if (kd != 0) ghost_signal(a, action, a.inter[lane], kd, false, sg, ds, ai);
        const float s2 = soft_switch(sg - 0.5f, kSigK);
        const float fr = s2 * grn_r + (1.0f - s2) * 1.0f, fu = s2 * grn_u + (1.0f - s2) * 0.0f;
        float g_fr = gg_r + g_own_in[2 * lane], g_fu = g_own_in[2 * lane + 1];      // the blended ghost is also the stored one
        glue_y_bwd(fr, fu, um, gg_y, g_fr, g_fu);
        float go_r = 0.f, go_u = 0.f;
        if (src >= 0) {
            add_r = g_fr * s2;
            glue_u_bwd(r[first], y[first], um, g_fu * s2, add_r, add_y);
        } else {
            go_r = g_fr * s2; go_u = g_fu * s2;
        }
        g_own_out[2 * lane] = go_r; g_own_out[2 * lane + 1] = go_u;
        if (kd != 0) {
            const float g_s2 = g_fr * (grn_r - 1.0f) + g_fu * grn_u;
            a_val = g_s2 * soft_switch_grad(sg - 0.5f, kSigK) * ds; a_key = ai;
        }
    }
    float *sl = slot + (size_t)j * 4;
    sl[0] = add_r; sl[1] = add_y; sl[2] = a_val; sl[3] = (float)a_key;
}

// adjoint, part 2 (thread = lane m): the lane's edge cells take what the ghosts that looked at them left, in a fixed order
// (downstream lanes ascending for the last cell, upstream lanes ascending for the first; no atomics: results repeat bit for
// bit); threads q < sq also sum their intersection's action partials in (lane, side) order.  g_r, g_y [C] and g_action [A]
// are accumulated into.
__global__ void net_ghosts_gather_kernel(NetStepArgs a, const int32_t *__restrict__ nxt_ptr, const int32_t *__restrict__ nxt_idx,
                                         const int32_t *__restrict__ prv_ptr, const int32_t *__restrict__ prv_idx,
                                         const int32_t *__restrict__ inter_ptr, const int32_t *__restrict__ inter_idx,
                                         const float *__restrict__ slot, float *__restrict__ g_r, float *__restrict__ g_y,
                                         float *__restrict__ g_action) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < a.L) {
        const int first = a.off[m], last = first + a.ncell[m] - 1;
        float vr = 0.f, vy = 0.f;
        for (int e = nxt_ptr[m]; e < nxt_ptr[m + 1]; ++e) {
            const int b = nxt_idx[e];
            if (a.left_src[b] == m) { vr += slot[(size_t)(2 * b) * 4]; vy += slot[(size_t)(2 * b) * 4 + 1]; }
        }
        g_r[last] += vr; g_y[last] += vy;
        vr = 0.f; vy = 0.f;
        for (int e = prv_ptr[m]; e < prv_ptr[m + 1]; ++e) {
            const int b = prv_idx[e];
            if (a.right_src[b] == m) { vr += slot[(size_t)(2 * b + 1) * 4]; vy += slot[(size_t)(2 * b + 1) * 4 + 1]; }
        }
        g_r[first] += vr; g_y[first] += vy;
    }
    if (m < a.sq) {
        double v = 0.; int key = -1;
        for (int k = inter_ptr[m]; k < inter_ptr[m + 1]; ++k) {
            const float *sl = slot + (size_t)inter_idx[k] * 4;
            if (sl[3] >= 0.f) { v += (double)sl[2]; key = (int)sl[3]; }
        }
        if (key >= 0) g_action[key] += (float)v;
    }
}

static inline bool netstep_ok(const dhts_net_desc *d, const dhts_net_tables *t, int step) {
    return d && t && d->n_lanes > 0 && d->n_cells > 0 && d->n_steps > 0 && step >= 0 && step < d->n_steps && d->n_inter_sq > 0 &&
           d->frames_per_phase > 0 && d->n_action >= d->n_inter_sq && t->lane_ncell && t->lane_off && t->sig_kind && t->inter &&
           t->left_src && t->left_gate && t->right_src && t->schedule;
}
static inline NetStepArgs netstep_args(const dhts_net_desc *d, const dhts_net_tables *t, int step) {
    NetStepArgs a;
    a.L = d->n_lanes; a.sq = d->n_inter_sq; a.F = d->frames_per_phase; a.n_action = d->n_action;
    a.phase_raw = step / d->frames_per_phase; a.frame = step % d->frames_per_phase;
    a.um = (float)d->u_max; a.um_d = d->u_max;
    a.ncell = t->lane_ncell; a.off = t->lane_off; a.kind = t->sig_kind; a.inter = t->inter;
    const size_t o = (size_t)step * d->n_lanes;
    a.left_src = t->left_src + o; a.left_gate = t->left_gate + o; a.right_src = t->right_src + o; a.schedule = t->schedule + o;
    return a;
}

}  // namespace dhts

using namespace dhts;

extern "C" {

int dhts_net_ghosts_fwd(const dhts_net_desc *d, const dhts_net_tables *t, int step, int hard, const float *action, const float *r,
                        const float *u, const float *own_in, float *own_out, float *ghost, void *stream) {
    if (!netstep_ok(d, t, step) || !action || !r || !u || !own_in || !own_out || !ghost) return DHTS_E_INVALID;
    const int n = 2 * d->n_lanes;
    net_ghosts_fwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(netstep_args(d, t, step), hard, action, r, u, own_in, own_out, ghost);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

int dhts_net_ghosts_bwd(const dhts_net_desc *d, const dhts_net_tables *t, const int32_t *inter_ptr, const int32_t *inter_idx, int step,
                        const float *action, const float *r, const float *y, const float *u, const float *own_in,
                        const double *g_ghost, const float *g_own_in, float *g_own_out, float *g_r, float *g_y, float *g_action,
                        float *scratch, void *stream) {
    if (!netstep_ok(d, t, step) || !t->nxt_ptr || !t->nxt_idx || !t->prv_ptr || !t->prv_idx || !inter_ptr || !inter_idx || !action ||
        !r || !y || !u || !own_in || !g_ghost || !g_own_in || !g_own_out || !g_r || !g_y || !g_action || !scratch)
        return DHTS_E_INVALID;
    const NetStepArgs a = netstep_args(d, t, step);
    const int n = 2 * d->n_lanes;
    net_ghosts_bwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, action, r, y, u, own_in, g_ghost, g_own_in, g_own_out, scratch);
    const int m = d->n_lanes > d->n_inter_sq ? d->n_lanes : d->n_inter_sq;
    net_ghosts_gather_kernel<<<(m + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, t->nxt_ptr, t->nxt_idx, t->prv_ptr, t->prv_idx, inter_ptr,
                                                                              inter_idx, scratch, g_r, g_y, g_action);
    return hipGetLastError() == hipSuccess ? DHTS_OK : DHTS_E_LAUNCH;
}

}  // extern "C"
