/*
 * dhts.h -- C ABI of libdhts.so: the MI355X (gfx950) differentiable traffic stepper.
 *
 * The reference (SonSang/diff-hybrid-traffic-sim) has NO FFI: its hot path sits behind two Python
 * torch.autograd.Function operators,
 *     dMacroForwardLayer   road/lane/dmacro_lane.py:234-309   (ARZ cell stencil, forward + Jacobian tape)
 *     dMicroForwardLayer   road/lane/dmicro_lane.py:228-298   (IDM car-following ODE, forward + tape)
 * each called once per lane per step from RoadNetwork.forward (road/network/road_network.py:99-101).
 * This header is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md): plain
 * pointers and sizes, no torch types.  Every entry point cites the reference code it replaces.
 *
 * Conventions
 *   - All array pointers are DEVICE pointers owned by the caller (hipMalloc / torch CUDA tensors),
 *     float32 unless stated.  Nothing is allocated, freed or synchronised inside a call; work is
 *     enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - Return value: DHTS_OK, or a negative DHTS_E_* for a host-side failure (bad argument, launch
 *     error).  Simulation faults the reference reports with `assert` (CFL violation
 *     _macro_lane.py:141-146, vehicle collision _micro_lane.py:151-162) are written to the optional
 *     device-side sticky record `dhts_error*` (first fault wins) and read back by the caller after the
 *     rollout; the rollout itself continues the way the reference's arithmetic would.
 *   - Layouts: state planes are [lane][cell] (cell fastest).  The Jacobian tapes are internal to this
 *     library but documented so they can be inspected.  There are two formats per model:
 *       single-step operators (dhts_macro_step_*, dhts_micro_step_*) keep the reference's blocks,
 *         macro: [lane][3][Np][4] float32, Np = dhts_padded(N) (multiple of 64); plane k = 0/1/2 holds
 *                d(next cell a)/d(cell a-1 / a / a+1) as row-major 2x2 in (r, y) -- the reference's
 *                dqs[a][k] (dmacro_lane.py:50-56) with the cell index moved inside for coalescing;
 *         micro: [lane][2][Vp][4] float32; plane 0 = dEgo, plane 1 = dLeading (dmicro_lane.py:48-54);
 *       rollouts (dhts_macro_rollout_*, dhts_micro_rollout_*) keep COMPACT tapes the reverse sweep
 *         rebuilds those blocks from: see dhts_macro_tape_bytes and dhts_micro_tape_bytes below;
 *         dhts_macro_tape_expand writes a rollout's blocks out in the single-step layout.
 */
#ifndef DHTS_H
#define DHTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DHTS_VERSION 100 /* 0.1.0 */

#define DHTS_OK 0
#define DHTS_E_INVALID (-1)   /* bad argument (NULL where required, non-positive size, unsupported size) */
#define DHTS_E_LAUNCH (-2)    /* HIP launch / runtime error */
#define DHTS_E_NO_DEVICE (-3) /* no gfx950 device visible */

/* fault codes in dhts_error.code */
#define DHTS_FAULT_NONE 0
#define DHTS_FAULT_CFL 1       /* dt >= dx / max(|speed|, 1e-5) at an interface  (_macro_lane.py:141-146) */
#define DHTS_FAULT_COLLISION 2 /* gap to the leader < 0                           (_micro_lane.py:151-162) */
#define DHTS_FAULT_NAN 3       /* non-finite cotangent in the reverse sweep       (dmacro_lane.py:308)     */
#define DHTS_FAULT_CAPACITY 4  /* hybrid network: record stream / vehicle slots / lane list / route table exhausted */

typedef struct dhts_error {
    int32_t code;  /* DHTS_FAULT_*; 0 = no fault.  Caller zeroes it before the first call. */
    int32_t step;  /* step index within the rollout */
    int32_t lane;  /* lane index */
    int32_t index; /* interface / vehicle index within the lane */
} dhts_error;

/* ---- macro (ARZ) ------------------------------------------------------------------------------- */
typedef struct dhts_macro_desc {
    int32_t n_lanes;  /* L: independent lanes in the batch */
    int32_t n_cells;  /* N: cells per lane (1 .. DHTS_MACRO_MAX_CELLS) */
    double dt;        /* delta_time */
    double dx;        /* cell_length (MacroLane.cell_length, _macro_lane.py:44) */
    double u_max;     /* speed_limit */
} dhts_macro_desc;

#define DHTS_MACRO_MAX_CELLS 4000

int dhts_version(void);
/* number of visible gfx950 devices, or DHTS_E_NO_DEVICE */
int dhts_device_count(void);
/* tuning knobs (process-wide, not part of the numerical contract).
 * DHTS_OPT_MACRO_FWD_WAVES: wavefronts per traffic lane in the macro forward kernel, 1..16 (the single-step operator's
 * kernel uses at most 8); 0 = heuristic. */
#define DHTS_OPT_MACRO_FWD_WAVES 1
/* DHTS_OPT_MICRO_FWD_WAVES: wavefronts per traffic lane in the micro forward kernel, 1, 2 or 4; 0 = heuristic. */
#define DHTS_OPT_MICRO_FWD_WAVES 2
/* DHTS_OPT_MACRO_FWD_VARIANT: kernel behind dhts_macro_rollout_fwd: 0 = two-phase kernels (trivial interfaces solved in
 * place, the others queued and solved compacted; full lanes of 128 W cells without a history take the pair kernel -- a thread owns
 * two adjacent cells and their right interfaces), 1 = the one-phase kernel of dhts_macro_step_fwd (every interface an
 * exception), 2 = the two-phase LANE kernel for every shape (one cell and its left interface per thread-pass, one lane per
 * workgroup: what ragged lanes and state histories take anyway; selectable so that tests can hold the pair kernel to it).
 * Same tape format, same results. */
#define DHTS_OPT_MACRO_FWD_VARIANT 3
/* DHTS_OPT_MACRO_FWD_GROUP: traffic lanes per workgroup of the pair kernel: 0 = heuristic (default: 4 for lanes of up to three
 * wavefronts, 1 for lanes of four -- BASELINE config 2 --, 2 above), or 1, 2, 4 (taken where the lanes divide, the launch keeps
 * >= 256 workgroups and the group fits a workgroup and the LDS).  Same results, same tape (dhts_macro_rollout_plan plan[7] says
 * what a shape gets). */
#define DHTS_OPT_MACRO_FWD_GROUP 4
/* DHTS_OPT_MACRO_FWD_ROTATE: 1 (default) = the workgroups of the second half of the pair kernel's grid take turns with the others at
 * issue priority 1, eight steps at a time; 0 = nobody does.  The rotation rests on an OBSERVATION about the dispatcher (workgroups b
 * and b + grid / 2 share a compute unit at two workgroups per unit, and the one dispatched second loses the arbitration), is used for
 * speed only and changes no result; bench.py times both settings during its warm-up and reports which one the box prefers. */
#define DHTS_OPT_MACRO_FWD_ROTATE 5
/* DHTS_OPT_NETSTEP_LDS_KB: KB of LDS the persistent kernels of dhts_netstep_rollout_* may plan with for what they keep resident (static
 * tables + per-step rows + ghosts, the micro side's running state, state rows / cotangent planes: dropped in that order of priority when
 * they do not fit); 1 .. 158, 0 = default (158).  Chooses among instantiations that compute the same numbers -- the tests run them all. */
#define DHTS_OPT_NETSTEP_LDS_KB 6
/* DHTS_OPT_NETSTEP_BLOCK: threads per workgroup of the persistent kernels: 256, 512, 1024; 0 = heuristic.  Same results. */
#define DHTS_OPT_NETSTEP_BLOCK 7
/* DHTS_OPT_HYB_PACK: replicas per compute unit of the fused hybrid network kernels (dhts_net_hybrid_rollout_fwd / _bwd): 0 = one;
 * 1 = two whenever the plan fits (half the LDS per workgroup: a smaller record staging area, temporaries for the network's own
 * micro lanes only; the 128-register instantiation); 2 (default) = two when the batch has more replicas than the device has
 * compute units.  Same results bit for bit; the value must not change between a forward sweep and its reverse (the reverse
 * sweep raises DHTS_FAULT_CAPACITY with index -3 otherwise).  dhts_net_hybrid_plan tells what a launch would take. */
#define DHTS_OPT_HYB_PACK 8
/* DHTS_OPT_REWARD_CHAIN: 1 = every network rollout (dhts_net_macro_rollout_fwd / _eval, dhts_net_hybrid_rollout_fwd / _eval,
 * dhts_netstep_rollout_fwd) finishes with the reward as ItscpEnv._reward forms it (example/control/itscp/_env.py:770-797): ONE
 * running float32 sum over lanes (outermost) and steps -- L x T dependent additions, one wavefront per replica, ~0.4 ms at config 4
 * -- instead of the kernels' own order (a lane's steps first, then the lanes' subtotals: the same real number, another
 * float32 rounding, 6e-6 relative at config 4).  An evaluation episode follows the reference's mixed chain (a Python float
 * until the first cell lane's tensor term joins it).  0 (default).  The queue terms and the gradient are the same either way. */
#define DHTS_OPT_REWARD_CHAIN 9
int dhts_set_option(int option, int value);
/* cells / vehicle slots rounded up to the tape's padded width (multiple of 64) */
int dhts_padded(int n);
/* bytes of Jacobian tape for T fused steps (dhts_macro_rollout_*).  The rollout tape holds, per (step, lane) row, what the
 *   reverse sweep needs to form the interface products A_i = flux'(Q_0) dQ_0/dQ_L, B_i = flux'(Q_0) dQ_0/dQ_R (the two 2x2
 *   products of dMacroLane._backward, dmacro_lane.py:116-124) of every interface i = 0 .. n_cells.  ROW LAYOUT (three
 *   blocks, each rounded up to a whole number of 128-byte lines; tests/test_boundary.py parses the three lines below):
 *     S: float32 [n_cells][3]                     (fp[0], fp[2], fp[3]) of interface i < n_cells, valid where it is trivial
 *     H: uint32 cnt, uint32 0, uint16 idx[n_cells + 1]   idx[j] = the interface of exception j < cnt
 *     E: float32 [n_cells + 1][2][4]              (A, B) of exception j; only the first cnt entries are written and read
 *   Where Q_0 = Q_L (the trivial Riemann case) dQ_0/dQ_L = I and dQ_0/dQ_R = 0, so A_i is the flux Jacobian fp at Q_L
 *   (darz.py:217-233), whose entry [0][1] is the constant 1, and B_i = 0: three floats.  Every other interface -- Q_0 is not
 *   Q_L, or the forward kernel chose to solve it in full; interface n_cells always -- is an "exception" and keeps its two
 *   products in E, in the order the forward solved them (its S entry is then unspecified).  There is no bit mask: a reader
 *   marks the interfaces listed in idx[0 .. cnt) and treats the rest as trivial.  The reverse sweep forms the reference's
 *   cell blocks dqs[a][0] = c A_a, dqs[a][1] = I - c (A_{a+1} - B_a), dqs[a][2] = - c B_{a+1} (c = dt / dx;
 *   dmacro_lane.py:126-129) from the products with the same float32 operations, so the results are those of the 48-byte
 *   per-cell tape; dhts_macro_tape_expand writes them out.
 *   The row is sized for the worst case (every interface an exception); the bytes MOVED per row are 12 n_cells + 8 +
 *   2 cnt + 32 cnt rounded to lines (BASELINE config 2: cnt ~ 0.15 n_cells, 8.8 KB per 512-cell row against 24.6 KB of
 *   dqs blocks). */
size_t dhts_macro_tape_bytes(const dhts_macro_desc *d, int T);
/* the reference's blocks dqs[a][3][2][2] (dmacro_lane.py:56) of all T steps from a rollout tape, in the single-step operator's
 *   layout per step: dqs_out [T][lane][3][Np][4] float32 (T x dhts_macro_step_tape_bytes).  For tests and for callers that want
 *   the Jacobians themselves; the reverse sweep does not need it. */
int dhts_macro_tape_expand(const dhts_macro_desc *d, int T, const float *tape, float *dqs_out, void *stream);
/* bytes of the single-step operator's tape (dhts_macro_step_*): the reference's per-cell blocks
 *   [lane][3][Np][4] float32, Np = dhts_padded(n_cells): plane k holds dqs[a][k] of every cell a */
size_t dhts_macro_step_tape_bytes(const dhts_macro_desc *d);

/*
 * n independent cell interfaces: ARZ.riemann_solve (model/macro/_arz.py:212-332) + dARZ.compute_dLdR and
 * dARZ.flux_prime (model/macro/darz.py:194-233) + the two 2x2 products of dMacroLane._backward
 * (road/lane/dmacro_lane.py:126-129), exactly as the rollout kernel evaluates them per interface.
 *   in  [9][n] DOUBLE (SoA): rL yL uL ueqL rR yR uR ueqR u_max  (float32-valued state widened to double)
 *   variant 0 = production arithmetic, 1 = reference-order IEEE division / square root
 *   out case_ind [n] int32 (0 = Q_L, 1 = Q_M, 2 = Q_C); q0 [4][n] DOUBLE (r, y, u, u_eq of Q_0);
 *       flux [2][n] DOUBLE (r u, y u of Q_0); dL, dR, fp [4][n] float32 row-major 2x2; A = fp @ dL, B = fp @ dR [4][n];
 *       cfl_bad [n] int32 = the CFL assert of _macro_lane.py:141-146 would fire for (dt, dx);
 *       speed [2][n] DOUBLE or NULL = (speed0, speed1) as ARZ.riemann_solve returns them (_arz.py:316-332), reference-order
 *       arithmetic in both variants (the rollout kernels only test them against the CFL bound)
 */
int dhts_arz_interface_batch(int64_t n, int variant, const double *in, double dt, double dx, int32_t *case_ind, double *q0,
                             double *flux, float *dL, float *dR, float *fp, float *A, float *B, int32_t *cfl_bad, double *speed,
                             void *stream);

/* float32 glue of FullQ.set_r_u / FullQ.from_r_u (model/macro/_arz.py:73-86 with :121-138):
 * y = r * (u - u_eq(r)), u_eq = u_max * (1 - sqrt(max(r, 0) + 1e-5)); n elements. */
int dhts_macro_state_from_ru(int64_t n, double u_max, const float *r, const float *u, float *y, float *ueq, void *stream);
/* its adjoint as torch autograd evaluates it: g_r += g_y * d y/d r, g_u = g_y * r   (g_r in/out, g_u out) */
int dhts_macro_state_from_ru_bwd(int64_t n, double u_max, const float *r, const float *u, const float *g_y,
                                 float *g_r, float *g_u, void *stream);
/* adjoint of the speed tap u = y / max(r, eps) + u_eq(max(r, eps)) of FullQ.set_r_y (_arz.py:88-92,126-131):
 * g_r += g_u * d u/d r, g_y += g_u * d u/d y   (both in/out) */
int dhts_macro_u_tap_bwd(int64_t n, double u_max, const float *r, const float *y, const float *g_u,
                         float *g_r, float *g_y, void *stream);

/*
 * T fused steps of L independent lanes: replaces T x L calls of dMacroForwardLayer.forward
 * (dmacro_lane.py:236-275 = MacroLane.forward _macro_lane.py:83-146 + dMacroLane._backward :96-132 +
 * the float32 u/u_eq glue of set_next_state_vector_y _macro_lane.py:282-299).
 *   in : r, y, u, ueq [L][N]; ghost [L][2][4] = per lane (left, right) x (r, y, u, ueq), constant over the
 *        rollout (RoadNetwork.setup_macro_boundary for a lane without neighbours, road_network.py:299-387)
 *   out: r_out, y_out, u_out, ueq_out [L][N] (may alias the inputs);
 *        tape (dhts_macro_tape_bytes) or NULL for a non-differentiable run;
 *        hist [T][L][3][N] = (r, y, u) after every step, or NULL
 */
int dhts_macro_rollout_fwd(const dhts_macro_desc *d, int T,
                           const float *r, const float *y, const float *u, const float *ueq, const float *ghost,
                           float *r_out, float *y_out, float *u_out, float *ueq_out,
                           float *tape, float *hist, dhts_error *err, void *stream);
/*
 * Reverse sweep over the tape: replaces T x L calls of dMacroForwardLayer.backward (dmacro_lane.py:277-309).
 *   in : g_r, g_y [L][N] cotangent of the final (r, y); g_hist [T][L][2][N] optional cotangent of the
 *        (r, y) after every step (NULL = none)
 *   out: g_r_out, g_y_out [L][N] cotangent of the initial (r, y) (may alias the inputs);
 *        g_ghost [L][2][2] DOUBLE = per lane (left, right) x (r, y): sum over steps of the cotangent that
 *        reaches the ghost cells (grad_ry[0], grad_ry[-1] of dmacro_lane.py:302-303)
 */
int dhts_macro_rollout_bwd(const dhts_macro_desc *d, int T, const float *tape,
                           const float *g_r, const float *g_y, const float *g_hist,
                           float *g_r_out, float *g_y_out, double *g_ghost, dhts_error *err, void *stream);

/* Which kernel instantiations the two calls above launch for this shape and the current options (answered by the functions
 * the launches themselves call; tests pin the benchmarked instantiations with it):
 *   plan[0] forward kernel: 0 = two-phase lane kernel, 1 = one-phase, 2 = two-phase pair kernel
 *   plan[1] wavefronts per lane     plan[2] 64-cell passes per wavefront (pair kernel: two adjacent cells per thread)
 *   plan[3] 1 = the full-lane, history-free instantiation (n_cells = 64 x passes x wavefronts and hist == NULL)
 *   plan[4] reverse kernel: 1 = pipelined one-cell-per-thread, 2 = pipelined two-cells-per-thread, 0 = general     plan[5] its block size
 *   plan[6] 1 = per-step cotangents / history requested (want_hist)
 *   plan[7] traffic lanes per workgroup (the pair kernel: DHTS_OPT_MACRO_FWD_GROUP; 1 otherwise) */
int dhts_macro_rollout_plan(const dhts_macro_desc *d, int T, int want_hist, int32_t plan[8]);

/* One step = the drop-in for a batch of dMacroForwardLayer.forward / .backward calls (T = 1 of the above;
 * tape is one step's worth).
 * A ghost cell of Python floats: the reference's Riemann solve reads a boundary cell that holds plain floats in double
 * (dMacroLane.decell leaves them alone); the one such cell of the itscp networks is a SOURCE lane's upstream ghost (_simulator.py:68-71:
 * r = the inflow of the schedule, u = u_eq(r), hence y = 0).  The step operator takes it as the LEFT quad {NaN, 0, low 32 bits of the
 * double r, high 32 bits} (bit patterns in the float slots; csrc/arz_device.hpp::ghost_source_pack): a NaN density marks it, u = u_eq(r)
 * is evaluated in double.  dhts_net_ghosts_fwd writes source lanes that way; the cotangent of such a ghost is of no use.  (The T-step
 * rollouts of straight lanes above take float32 ghosts only.) */
int dhts_macro_step_fwd(const dhts_macro_desc *d,
                        const float *r, const float *y, const float *u, const float *ueq, const float *ghost,
                        float *r_out, float *y_out, float *u_out, float *ueq_out,
                        float *tape, dhts_error *err, void *stream);
int dhts_macro_step_bwd(const dhts_macro_desc *d, const float *tape, const float *g_r, const float *g_y,
                        float *g_r_out, float *g_y_out, double *g_ghost, dhts_error *err, void *stream);

/* ---- micro (IDM) ------------------------------------------------------------------------------- */
typedef struct dhts_micro_desc {
    int32_t n_lanes;   /* L */
    int32_t capacity;  /* V: vehicle slots per lane (1 .. DHTS_MICRO_MAX_VEHICLES); slot i follows slot i+1,
                          head = slot count-1 (MicroLane.curr_vehicle order, _micro_lane.py:31-34) */
    double dt;
} dhts_micro_desc;

#define DHTS_MICRO_MAX_VEHICLES 1024
#define DHTS_MICRO_NPARAM 6 /* accel_max, accel_pref, target_speed, min_space, time_pref, length (micro_vehicle.py:21-28) */

/* bytes of Jacobian tape for T fused steps (dhts_micro_rollout_*): [step][lane][Vp][3] float32 = (dEgo[1][0], dEgo[1][1],
 * dLeading[1][1]) of every vehicle (12 B per vehicle-step).  The first rows of both blocks are the constants [1, dt] and
 * [0, 0], and dLeading[1][0] = -dEgo[1][0] bit for bit (didm.py:38-103); the reverse sweep re-inserts them, so results equal
 * those of the 32-byte dqs[a][2][2][2] tape */
size_t dhts_micro_tape_bytes(const dhts_micro_desc *d, int T);
/* bytes of the single-step operator's tape (dhts_micro_step_*): [lane][2][Vp][4] float32, plane k = dqs[a][k] */
size_t dhts_micro_step_tape_bytes(const dhts_micro_desc *d);

/*
 * n independent vehicles: IDM.compute_acceleration (model/micro/_idm.py:6-50) + one explicit-Euler step
 * (_micro_lane.py:182-183) + dIDM.compute_dEgo / compute_dLeading (model/micro/didm.py:13-103).
 *   variant 0 = production arithmetic, 1 = reference-order IEEE division / square root
 *   in  [9][n] DOUBLE (SoA): a_max a_pref v v_target position_delta speed_delta min_space time_pref delta_time
 *   out next_pv [2][n] DOUBLE: (0 + dt v, v + dt acc) rounded to float32; dEgo, dLeading [4][n] float32;
 *       collided [n] int32 (position_delta < 0); acc_sstar [2][n] DOUBLE = (acceleration, clipped optimal spacing);
 *       clips [2][n] int32 = (clipped_acceleration, clipped_optimal_spacing)
 */
int dhts_idm_batch(int64_t n, int variant, const double *in, double *next_pv, float *dEgo, float *dLeading, int32_t *collided,
                   double *acc_sstar, int32_t *clips, void *stream);

/* The Jacobians alone, from the caller's optimal spacing and clip flags: dIDM.compute_dEgo / compute_dLeading as the reference
 * declares them (model/micro/didm.py:13-103; dMicroLane._backward passes the flags of the forward pass -- derived from the gap
 * clamped to 1e-5 -- beside the UN-clamped gap, dmicro_lane.py:97).
 *   in  [12][n] DOUBLE (SoA): a_max a_pref v v_target position_delta speed_delta min_space time_pref optimal_spacing delta_time
 *       clipped_acceleration clipped_optimal_spacing (0 / 1)
 *   out dEgo, dLeading [4][n] float32 */
int dhts_idm_jac_batch(int64_t n, const double *in, float *dEgo, float *dLeading, void *stream);

/*
 * T fused steps of L independent lanes: replaces T x L calls of dMicroForwardLayer.forward
 * (dmicro_lane.py:230-269 = MicroLane.forward _micro_lane.py:131-214 + dMicroLane._backward :87-127).
 *   in : p, v [L][V]; count [L] int32 vehicles per lane or NULL (= V everywhere);
 *        params [6][L][V] DOUBLE (the reference keeps them as Python floats);
 *        head [L][2] DOUBLE = (head_position_delta, head_speed_delta) per lane, constant over the rollout
 *        (defaults 1000 / 0, _micro_lane.py:14-15)
 *   out: p_out, v_out [L][V]; tape or NULL; hist [T][L][2][V] = (p, v) after every step, or NULL
 */
int dhts_micro_rollout_fwd(const dhts_micro_desc *d, int T,
                           const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                           float *p_out, float *v_out, float *tape, float *hist, dhts_error *err, void *stream);
/*
 * Reverse sweep: replaces T x L calls of dMicroForwardLayer.backward (dmicro_lane.py:271-298), including the
 * virtual-leader slot of dMicroLane.vectorize_input (:130-153) whose cotangent returns to the head vehicle and
 * to (head_position_delta, head_speed_delta).
 *   in : g_p, g_v [L][V]; g_hist [T][L][2][V] optional
 *   out: g_p_out, g_v_out [L][V]; g_head [L][2] DOUBLE = cotangent of (head_position_delta, head_speed_delta)
 */
int dhts_micro_rollout_bwd(const dhts_micro_desc *d, int T, const float *tape, const int32_t *count,
                           const float *g_p, const float *g_v, const float *g_hist,
                           float *g_p_out, float *g_v_out, double *g_head, dhts_error *err, void *stream);

/* Which kernel instantiations the two calls above launch (see dhts_macro_rollout_plan):
 *   plan[0] forward wavefronts per lane (1, 2, 4)     plan[1] passes per thread (the literal K: 1, 2, 4, 8, 16)
 *   plan[2] 1 = the full-lane instantiation (count == NULL and capacity = 64 x wavefronts x passes)
 *   plan[3] reverse sweep: 1 = the prefetched one-vehicle-per-thread path, 0 = the strided loop     plan[4] its block size */
int dhts_micro_rollout_plan(const dhts_micro_desc *d, int T, int has_count, int32_t plan[8]);

int dhts_micro_step_fwd(const dhts_micro_desc *d,
                        const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                        float *p_out, float *v_out, float *tape, dhts_error *err, void *stream);
/* The same step in the float32 TENSOR ladder: what the reference's plain MicroLane (road/lane/_micro_lane.py:131-214 over
 * model/micro/_idm.py:6-50) computes when its vehicle states are torch tensors -- every operation rounds to float32, Python float
 * operands are cast first, pow(x, 2.0) = x * x, pow(x, 4.0) = powf.  itscp `micro` mode steps its lanes that way in differentiable
 * episodes (example/control/itscp/_env.py:484-498).  Same arguments, same tape (the analytic Jacobian blocks at the same operands:
 * what autograd differentiates), same reverse operator dhts_micro_step_bwd. */
int dhts_micro_step_fwd_tensor(const dhts_micro_desc *d,
                               const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                               float *p_out, float *v_out, float *tape, dhts_error *err, void *stream);
/* The same step when only the lane's HEAD GAP is a float32 tensor: dMicroLane.detach_vehicle (road/lane/dmicro_lane.py:228-250) turns the
 * vehicles' states into Python floats but leaves head_position_delta / head_speed_delta alone, so with a tensor gap (a differentiable
 * itscp hybrid episode: the signal blend of example/control/itscp/_simulator.py:260-263; a plain network: a leader in sight) the head
 * vehicle's IDM.compute_acceleration and Euler step run in mixed arithmetic -- double where two Python floats meet, float32 where the
 * tensor is involved -- and the followers' in double as in dhts_micro_step_fwd.  head [L][2] double holds the tensor's float32 values.
 * Same arguments, same tape, same reverse operator. */
int dhts_micro_step_fwd_tensor_head(const dhts_micro_desc *d,
                                    const float *p, const float *v, const int32_t *count, const double *params, const double *head,
                                    float *p_out, float *v_out, float *tape, dhts_error *err, void *stream);
/* The single-step reverse keeps the operator form of dMicroForwardLayer.backward: the cotangent of the virtual
 * leader slot is NOT folded back into the head vehicle; g_head [L][2] DOUBLE returns it raw as (g_p[V], g_s[V]). */
int dhts_micro_step_bwd(const dhts_micro_desc *d, const float *tape, const int32_t *count,
                        const float *g_p, const float *g_v,
                        float *g_p_out, float *g_v_out, double *g_head, dhts_error *err, void *stream);


/* ---- macro road network with differentiable signals (itscp `macro` mode) -------------------------------------------
 * One fused rollout of R independent replicas of a signalised network of ARZ lanes: replaces, per replica, what
 * ItscpEnv._simulate does with ItscpRoadNetwork.forward (example/control/itscp/_env.py:620-768, 885-962;
 * _simulator.py:56-137; road/network/road_network.py:79-111, 299-387) and ItscpEnv._reward (_env.py:770-797):
 * per step  signals from the action -> ghost cells of every lane from its neighbours' time-n edge cells, blended between
 * green and red values -> one ARZ step per lane -> queue-length loss (running-mean-scaled sigmoid, _env.py:586-618).
 * Lanes start empty.  Tables (device pointers) are built on the host (dhts/network.py):
 *   lane_ncell, lane_off, sig_kind (0 always green, 1 west-east phase, 2 north-south phase), inter [L] int32; lane_dx [L]
 *   DOUBLE; left_src / left_gate / right_src [T][L] int32 and schedule [T][L] DOUBLE, per replica when replica_stride
 *   (elements between replicas) is non-zero, else shared.  The loss' RunningMean(100 000) window slides once T * n_cells exceeds it
 *   (the samples that leave are re-read from the state history).  Limits: n_cells + n_lanes <= 1024 (one workgroup per replica), n_action <= 1024, at most 4 upstream / 4 downstream lanes per lane.
 */
typedef struct dhts_net_desc {
    int32_t n_replicas, n_lanes, n_cells, n_steps, n_inter_sq, frames_per_phase, n_action;
    double dt, u_max, static_speed, vehicle_length;
} dhts_net_desc;
typedef struct dhts_net_tables {
    const int32_t *lane_ncell, *lane_off, *sig_kind, *inter;
    const double *lane_dx;
    const int32_t *left_src, *left_gate, *right_src;
    const double *schedule;
    int64_t replica_stride;
    /* static adjacency in CSR form, neighbour ids ascending: next lanes of lane l = nxt_idx[nxt_ptr[l] .. nxt_ptr[l+1]),
     * previous lanes likewise (used by the reverse sweep to route ghost cotangents without searching) */
    const int32_t *nxt_ptr, *nxt_idx, *prv_ptr, *prv_idx;
    int32_t n_edges;
} dhts_net_tables;
size_t dhts_net_macro_hist_bytes(const dhts_net_desc *d);   /* state history [R][T+1][4][C] float32 */
size_t dhts_net_macro_tape_bytes(const dhts_net_desc *d);   /* Jacobian tape [R][T][3][Cp][4] float32 */
/* action [R][n_action]; out: hist, tape, kc [R][T][C] (loss sigmoid constants), queue [R][T][L] (loss terms q^2 dt),
 * reward [R] float32 = - sum of the queue terms; workspace [R][T][2 L] float32 (kept for the reverse sweep) */
int dhts_net_macro_rollout_fwd(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, float *hist, float *tape,
                               float *kc, float *queue, float *reward, float *workspace, dhts_error *err, void *stream);
/* An EVALUATION episode of the same R replicas: ItscpEnv.step(action, False), what Trainer.evaluate runs every num_eval_epoch
 * epochs (example/control/trainer.py:73-75, 94-142).  The lanes step exactly as above; the thresholds are hard: signals
 * float(a > progress) / float(progress > a) (_env.py:928-960), downstream ghost float(signal > 0.5) (_simulator.py:128-137),
 * is_static = 1.0 if speed < static_speed else 0.0 without a running mean (_env.py:607-617).  Nothing is kept for a reverse
 * sweep.  out: queue [R][T][L], reward [R]. */
int dhts_net_macro_rollout_eval(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, float *queue, float *reward,
                                dhts_error *err, void *stream);
/* hist, tape, kc, queue, workspace from the forward; g_reward [R] (NULL = ones) -> g_action [R][n_action] */
int dhts_net_macro_rollout_bwd(const dhts_net_desc *d, const dhts_net_tables *t, const float *action, const float *hist,
                               const float *tape, const float *kc, const float *queue, const float *g_reward, float *g_action,
                               const float *workspace, dhts_error *err, void *stream);

/* ---- Macro road networks of ANY size, step by step (round 4) ---------------------------------------------------------------
 * The rollouts above keep a replica in one workgroup (cells + lanes <= 1024); the reference builds grids of any size
 * (example/control/itscp/_env.py:221-439).  A larger network steps as RoadNetwork.forward does (road_network.py:57-111), one
 * operator call per step for ALL lanes: dhts_net_ghosts_fwd turns the state before step `step` into every lane's two ghost
 * cells -- the connected neighbour's edge cell, the inflow schedule of a source lane or the stored downstream ghost of a sink
 * lane, blended between green and red by the signals of that step (ItscpRoadNetwork.setup_macro_boundary,
 * _simulator.py:42-60; signals _env.py:885-962) -- in the layout dhts_macro_step_fwd takes its ghosts in, the lanes then step
 * as batches of that operator (lanes of equal cells and cell length), and dhts_net_ghosts_bwd routes the operator's ghost
 * cotangents (dhts_macro_step_bwd's g_ghost) back: to the neighbours' edge cells (r, y), to the stored ghosts and to the action.
 * d / t as for dhts_net_macro_rollout_fwd with n_replicas = 1 (lane_off / lane_ncell give the edge cells; rows `step` of
 * left_src / left_gate / right_src / schedule are read).  r, y, u [C] = the state before the step (u = u(r, y) as the operator
 * left it); own_in / own_out [L][2] = the lanes' stored downstream ghosts (r, u) before / after (initially (0, u_max));
 * ghost [L][2][4] = (left, right) x (r, y, u, u_eq), a source lane's left quad in the double encoding dhts_macro_step_fwd documents;
 * hard != 0 = an evaluation episode's thresholds.
 * _bwd: g_ghost [L][2][2] double (d / d ghost (r, y)), g_own_in [L][2] = cotangent of own_out; g_own_out [L][2] is written;
 * g_r, g_y [C] and g_action [n_action] are ACCUMULATED into (edge cells in a fixed order: bit-repeatable); inter_ptr [sq + 1],
 * inter_idx = the ghost slots (2 lane + side) of every intersection in ascending order; scratch [L][2][4] float32. */
int dhts_net_ghosts_fwd(const dhts_net_desc *d, const dhts_net_tables *t, int step, int hard, const float *action, const float *r,
                        const float *u, const float *own_in, float *own_out, float *ghost, void *stream);
int dhts_net_ghosts_bwd(const dhts_net_desc *d, const dhts_net_tables *t, const int32_t *inter_ptr, const int32_t *inter_idx, int step,
                        const float *action, const float *r, const float *y, const float *u, const float *own_in,
                        const double *g_ghost, const float *g_own_in, float *g_own_out, float *g_r, float *g_y, float *g_action,
                        float *scratch, void *stream);

/* ---- HYBRID road network (itscp `hybrid` mode): macro lanes, micro lanes and the hand-offs between them -----------------
 * As the macro network rollout, with lanes that are either ARZ cell lanes or IDM vehicle lanes and, after every step,
 * the conversions of road/network/conversion.py:11-215 in lane-id order (RoadNetwork.conversion, road_network.py:113-170):
 *   macro -> micro  flux capacitor of the last cell (+= r u dt), spawn of a default vehicle (p = 0, v = u_last, ancillary
 *                   a = capacitor) once it holds one vehicle length and the successor has that much free space (:16-73)
 *   micro -> macro  the head vehicle past the lane end by more than its length is deposited into the successor's first
 *                   cells (density a/len * overlap/dx, straight-through clamp, cell speed = vehicle speed) (:76-171)
 *   micro -> micro  lane change at p >= L (:175-200);  micro -> none at p >= L (:203-215)
 * plus the head gap of every occupied micro lane: leader further along the route (road_network.py:429-580) blended with the
 * red-light stop line by position-weighted neighbour signals (_simulator.py:139-276), and the vehicles' terms of the queue
 * loss (_env.py:694-725).  Vehicles use MicroVehicle.default_micro_vehicle (micro_vehicle.py:31-72).
 * One workgroup per replica; the (few) scalar operations of the micro side run on an extra wavefront (one lane per micro
 * lane; the hand-off events serially) and are recorded, with their partial derivatives, on a per-replica record stream that
 * the reverse sweep replays backwards.
 * Extra tables (device pointers):
 *   lane_macro [L] int32 (1 = cells, 0 = vehicles; micro lanes have 0 cells), lane_len [L] DOUBLE,
 *   left_src -3 = own stored upstream ghost (single upstream lane is micro), right_src -1 also when the single downstream
 *   lane is micro, conv_next [T][L] int32 = the step's macro-route successor of a macro lane (per replica like left_src),
 *   routes [n_routes][route_stride] int32 (-1 padded), grouped by their first lane, and route_ptr [L + 1] int32 (rows
 *   route_ptr[m] .. route_ptr[m+1] start on micro lane m): the k-th vehicle spawned onto lane m takes row
 *   route_ptr[m] + k mod (rows of m).  The reference draws a route with np.random at spawn time (road_network.py:604-646);
 *   callers pre-draw them (dhts/network.py: group_routes keeps a recorded spawn order intact).
 * Limits: n_cells + n_lanes <= 960 (n_cells = 0 is allowed: an all-micro network), <= 64 micro lanes, <= 16 spawning lanes, <= lane_capacity vehicles per micro lane (16 unless asked for: below), <= 48 tape
 * records per micro lane and step (2 lane_capacity + 32 with a larger lane_capacity) -- fewer when more than ~20 micro lanes share the workgroup's LDS staging, e.g. 18 at 64
 * micro lanes beside 288 cells; a lane with k vehicles stages ~8 + 2 k (DHTS_FAULT_CAPACITY beyond), <= 128 vehicles per replica and episode, route_stride <= 32; records_per_step (average budget of the record stream,
 * 0 = 512).  loss_steps: only the first loss_steps steps enter reward_cut and the gradient (<= 0: all).
 */
typedef struct dhts_hybrid_tables {
    dhts_net_tables net;
    const int32_t *lane_macro;
    const double *lane_len;
    const int32_t *conv_next;
    const int32_t *routes;
    const int32_t *route_ptr;
    int32_t n_routes, route_stride, records_per_step, loss_steps;
    int32_t n_micro;            /* number of micro lanes (zeros of lane_macro); sizes the kernels' LDS staging */
    /* micro SOURCE lanes -- micro lanes without an upstream lane; itscp `micro` mode, where every lane is an IDM lane and
     * n_cells = 0 (ItscpRoadNetwork.setup_micro_boundary, _simulator.py:153-174): at the boundary of a step such a lane admits
     * a waiting vehicle (position 0, speed 0) when it has more than half a vehicle length of room at its entrance and the
     * host's draw np.random.random() is below the step's inflow schedule[t][lane].  The draws are data: `draws` is the stream
     * in call order (one draw per source lane with room, lanes in id order, steps in order), n_draws its length (a run that
     * needs more raises DHTS_FAULT_CAPACITY), draws_stride the elements between replicas (0 = shared).  The k-th vehicle admitted
     * to lane m takes route row route_ptr[m] + k (no wrap-around: the rows are the lane's waiting list in admission order, and an
     * exhausted list admits nobody).  lane_source [L] int32 (1 = source lane) or NULL = the network has none.
     * (Which arithmetic the IDM lanes step in is micro_tensor_ladder's business, below -- not inferred from lane_source.) */
    const int32_t *lane_source;
    const double *draws;
    int32_t n_draws;
    int64_t draws_stride;
    /* vehicles a micro lane holds at once: 0 (= 16), 16, 32, 64 or 128.  A per-launch LDS sizing like n_micro (4 bytes per slot and
     * micro lane, plus staging for one IDM record and one loss seed per vehicle and step, as far as the 160 KB go): the reference's
     * lanes are unbounded (_micro_lane.py:53-113), an episode that needs more than the launch was sized for ends in
     * DHTS_FAULT_CAPACITY and the caller retries with a larger value (ItscpEnv.step does) or runs lane by lane. */
    int32_t lane_capacity;
    /* 1 = the network's IDM lanes are the reference's PLAIN autodiff MicroLane objects on torch tensors -- itscp `micro` mode
     * (example/control/itscp/_env.py:484-498; road/lane/_micro_lane.py:131-214 evaluated by torch): in differentiable episodes the IDM
     * step follows that float32 TENSOR ladder operation by operation (csrc/idm_device.hpp idm_step_f32).  0 = dMicroLane lanes (every
     * other network, with or without source lanes): the analytic operator's float64 ladder (dmicro_lane.py:87-127), the head vehicle in
     * mixed arithmetic while its gap is a tensor.  The host says which (ItscpEnv's `micro` branch sets 1). */
    int32_t micro_tensor_ladder;
    /* The vehicles' IDM attributes, [n_routes][6] DOUBLE (device) beside the route table -- row i belongs to the vehicle that takes
     * route row i: (accel_max, accel_pref, target_speed, min_space, time_pref, length) as MicroVehicle holds them
     * (road/vehicle/micro_vehicle.py:5-28; random_micro_vehicle :75-121) -- or NULL: every vehicle is a
     * MicroVehicle.default_micro_vehicle(speed_limit) (:31-72), which is all the reference's network code ever builds
     * (road/network/conversion.py:51, road_network.py:582-591).  Rows are reused with their routes (the k-th vehicle spawned onto
     * lane m takes row route_ptr[m] + k mod rows of m).  length must equal the descriptor's vehicle_length (the hand-offs and the
     * loss use one length; both of the reference's factories give DEFAULT_VEHICLE_LENGTH). */
    const double *veh_params;
    /* Replicas per compute unit of the fused kernels for THIS set of tables: 0 = as DHTS_OPT_HYB_PACK says (default: two when the batch
     * has more replicas than the device has units), 1 = two whenever the plan fits, -1 = one.  A caller that met DHTS_FAULT_CAPACITY under
     * the packed plan (its record staging area holds about a third of the unpacked plan's records per lane and step: dhts_net_hybrid_plan plan[2]) retries with -1 (dhts.ops does). */
    int32_t two_per_cu;
} dhts_hybrid_tables;
size_t dhts_net_hybrid_workspace_bytes(const dhts_net_desc *d, const dhts_hybrid_tables *t);
/* What dhts_net_hybrid_rollout_fwd / _bwd would launch for (d, t) under the current DHTS_OPT_HYB_PACK (no device work): plan[0] = 1
 * when two replicas share a compute unit, [1] threads per workgroup, [2] records a micro lane can stage per step, [3] / [4]
 * bytes of LDS of the forward / reverse kernel, [5] lanes with a range of temporaries, [6] records a step may hold, [7] compute
 * units of the current device.  (No reference counterpart: the reference steps one environment in one Python thread,
 * example/control/trainer.py:168-204.) */
int dhts_net_hybrid_plan(const dhts_net_desc *d, const dhts_hybrid_tables *t, int32_t plan[8]);
/* bytes of the hybrid kernels' Jacobian tape: float32 [R][T][NIp][2][4], NIp = n_cells + n_lanes rounded up to 64: per interface
 * slot (a lane's n + 1 interfaces, lanes in id order, micro lanes own none) the two 2x2 products A = flux'(Q_0) dQ_0/dQ_L and
 * B = flux'(Q_0) dQ_0/dQ_R of dMacroLane._backward (dmacro_lane.py:116-124); the reverse sweep forms the cell blocks
 * dqs[a][0..2] (:126-129) from them.  (The macro-network kernels keep the blocks themselves: dhts_net_macro_tape_bytes.) */
size_t dhts_net_hybrid_tape_bytes(const dhts_net_desc *d);
/* hist / kc / queue / reward as in the macro rollout (kc rows of micro lanes do not exist: they have no cells); tape:
 * dhts_net_hybrid_tape_bytes;
 * counts [R][4] int32 = (vehicles spawned, vehicles deposited, records written, 0) */
int dhts_net_hybrid_rollout_fwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, const float *action, float *hist,
                                float *tape, float *kc, float *queue, float *reward, int32_t *counts, void *workspace,
                                dhts_error *err, void *stream);
/* An EVALUATION episode (see dhts_net_macro_rollout_eval) of the hybrid network: additionally a head vehicle takes the green
 * head gap when the signal of its own lane is >= 0.5 and the red-light gap otherwise (_simulator.py:208-232, 264-276: scores
 * 0 / 1 / 0, no running mean), and a vehicle is static when its speed is below static_speed (_env.py:709-717).  Steps, hand-offs
 * and capacities as in dhts_net_hybrid_rollout_fwd; no records, tape, history or workspace.  out: queue [R][T][L], reward [R],
 * counts [R][4] = (vehicles spawned, vehicles deposited, 0, 0). */
int dhts_net_hybrid_rollout_eval(const dhts_net_desc *d, const dhts_hybrid_tables *t, const float *action, float *queue,
                                 float *reward, int32_t *counts, dhts_error *err, void *stream);
int dhts_net_hybrid_rollout_bwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, const float *action, const float *hist,
                                const float *tape, const float *kc, const float *queue, const float *g_reward,
                                float *g_action, const void *workspace, dhts_error *err, void *stream);

/* ---- the same kernels for a PLAIN road network with given initial state: example/inverse/hybrid.py ----------------------------
 * RoadNetwork.forward of the reference without the itscp layer (road/network/road_network.py:79-111, 299-362, 429-580): no
 * signals and no inflow schedules -- a ghost is the connected macro lane's edge cell or the lane's own STORED ghost, unblended
 * (get_macro_boundary), a head vehicle's gap is the one to its leader along its route (setup_micro_boundary) -- lanes that start
 * from a GIVEN state, and taps on the FINAL state instead of the queue loss: what example/inverse/hybrid.py:37-146 and
 * _inverse.py:91-99, 185-242 run T x (3 operator calls + host conversions) for.
 *   plain       1 = the plain network above (tables: sig_kind all 0, left_src / right_src = neighbour lane or -3 / -1 for the
 *               stored ghost, schedule unused); 0 = the itscp semantics of dhts_net_hybrid_rollout_fwd with the extra inputs
 *   state0      [R][4][C] float32 (r, y, u, u_eq) of every cell at step 0 (NULL = empty road: r = y = 0, u = u_eq = u_max)
 *   ghost0      [R][L][4] float32 (r, u) of each lane's stored upstream ghost, (r, u) of its stored downstream ghost
 *               (NULL = (0, u_max) both: set_leftmost_cell / set_rightmost_cell defaults)
 *   veh_out     [R][128][4] float32 out: (lane id or -1 once it left the network, position, speed, ancillary a) of every
 *               vehicle in spawn order (rows beyond counts[0] are not written)
 *   events      [R][256][2] int32 out: (step, kind) of the hand-off events in order, kind 0 = macro -> micro spawn, 1 = micro ->
 *               macro deposit; counts[3] = their number (NULL = not kept)
 * Reverse: g_stateT [R][3][C] cotangent of the final (r, y, u) per cell and g_veh [R][128][2] cotangent of the final (position,
 * speed) of vehicle k (either may be NULL) enter beside g_reward (NULL = ones, as above; pass zeros for a pure state tap);
 * g_state0 [R][3][C] out = cotangent of the initial (r, y, u) -- u collects what reads the GIVEN speed at step 0 (ghosts of
 * neighbouring lanes, a flux capacitor); the initial y = r (u - u_eq(r)) is the caller's (dhts_macro_state_from_ru_bwd). */
typedef struct dhts_hybrid_state_io {
    int32_t plain;
    const float *state0;
    const float *ghost0;
    float *veh_out;
    int32_t *events;
} dhts_hybrid_state_io;
int dhts_net_hybrid_state_rollout_fwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, const dhts_hybrid_state_io *io,
                                      const float *action, float *hist, float *tape, float *kc, float *queue, float *reward,
                                      int32_t *counts, void *workspace, dhts_error *err, void *stream);
int dhts_net_hybrid_state_rollout_bwd(const dhts_net_desc *d, const dhts_hybrid_tables *t, int32_t plain, const float *action,
                                      const float *hist, const float *tape, const float *kc, const float *queue,
                                      const float *g_reward, const float *g_stateT, const float *g_veh, float *g_action,
                                      float *g_state0, const void *workspace, dhts_error *err, void *stream);

/* ---- Road networks of ANY size and ANY mix of ARZ / IDM lanes, step by step (round 5) ------------------------------------------
 * dhts_net_hybrid_rollout_* keep a replica in one workgroup (cells + lanes <= 960, <= 64 IDM lanes, <= 128 vehicles); the
 * reference's environment builds any grid in any mode (example/control/itscp/_env.py:221-506: `--lane_length`, `--n_lane`,
 * `--n_intersection` are free).  dhts_netstep_rollout_fwd / _bwd run such an episode step by step, every step a handful of
 * launches over flat device arrays, all T steps inside ONE call (no host round trip, no per-step Python):
 *   boundary   ghost cells of every ARZ lane (ItscpRoadNetwork.setup_macro_boundary, _simulator.py:56-137) | admission of
 *              waiting vehicles on micro source lanes (:153-174), head gap of every occupied IDM lane (:139-276 over
 *              RoadNetwork.setup_micro_boundary, road_network.py:429-580; forward-mode duals w.r.t. the head vehicle, its
 *              leader and the three signals it can see) and the IDM step of every vehicle (dMicroForwardLayer,
 *              dmicro_lane.py:230-269) -- one launch;
 *   lanes      dhts_macro_step_fwd once per group of ARZ lanes of equal (cells, cell length): dMacroForwardLayer for the batch;
 *   hand-offs  flux capacitors, then spawn / lane change / despawn / deposit in lane-id order (RoadNetwork.conversion,
 *              road_network.py:113-170; conversion.py:11-215) with an event list for the reverse sweep, then the queue loss of
 *              the committed state with its RunningMean(100 000) (_env.py:586-618, 664-742) -- one launch.
 * The reverse sweep undoes them newest first (loss taps -> events -> IDM / head gaps -> dhts_macro_step_bwd per group -> ghost
 * cotangents to the neighbours' edge cells, the stored ghosts and the action).  Nothing is fused across steps: a step costs its
 * launches (~2 + groups forward, ~3 + groups backward), which is what makes any size run; networks inside the limits above
 * are ~10 x faster through dhts_net_hybrid_rollout_*.
 * Tables: dhts_hybrid_tables in the network's own lane ids (conversions, loss samples and running means follow lane-id order),
 * EXCEPT that cells are stored GROUP-MAJOR: hyb.net.lane_off[l] is the first cell of lane l in an order where the lanes of a
 * group are contiguous (the batched operator works on the state arrays in place); lane_gpos[l] is the lane's row in that order
 * (ghost rows).  hyb.lane_capacity = vehicles a micro lane holds (any value 1 .. 1024; DHTS_FAULT_CAPACITY beyond), no limit on
 * the number of micro lanes, spawning lanes or vehicles.  n_replicas must be 1.
 *   hist   [T + 1][4][C] float32 out: (r, y, u, u_eq) of every cell before step t (row 0: the empty road)
 *   queue  [T][L] out; reward [2] out = (- sum of all queue terms, the same over the first loss_steps steps)
 *   counts [4] int32 out = (vehicles spawned or admitted, vehicles deposited, hand-off events, admission draws consumed)
 * hard != 0: an evaluation episode (hard thresholds, see dhts_net_hybrid_rollout_eval); nothing is kept for a reverse sweep.
 * _bwd: g_reward [1] or NULL (= 1) -> g_action [n_action] (gradient of reward[1]). */
typedef struct dhts_netstep_group {
    int32_t lane_pos0;   /* first row of the group in the group-major lane order */
    int32_t n_lanes;     /* lanes in the group */
    int32_t n_cells;     /* cells per lane */
    int32_t cell0;       /* first cell of the group */
    double dx;           /* cell length */
} dhts_netstep_group;
typedef struct dhts_netstep_tables {
    dhts_hybrid_tables hyb;
    const int32_t *lane_gpos;          /* [L] device: row of macro lane l in the group-major lane order (-1: micro lane) */
    const dhts_netstep_group *groups;  /* HOST pointer */
    int32_t n_groups;
    const int32_t *micro_lanes;        /* [hyb.n_micro] device: ids of the micro lanes, ascending */
    const int32_t *lane_mslot;         /* [L] device: position of lane l in micro_lanes (-1: macro lane) */
    const int32_t *cap_lanes;          /* [n_caps] device: macro lanes with a micro successor (flux capacitors), ascending */
    const int32_t *lane_cslot;         /* [L] device: position of lane l in cap_lanes (-1: none) */
    int32_t n_caps;
    const int32_t *inter_ptr, *inter_idx; /* device: ghost slots (2 lane + side) of every intersection, ascending ([sq + 1], [..]) */
    int32_t max_events;                /* capacity of the hand-off event list of an episode (0 = 8 per step) */
    /* persistent != 0: the PERSISTENT form -- one kernel per direction, one workgroup per replica, all T steps, the workgroup's
     * threads looping over the network's items with workgroup barriers where the stepwise form has kernel boundaries (same device
     * functions: same numbers).  ~30 us per step (forward + reverse) at 360 lanes / 2 124 cells against ~80 us of launches; the
     * one workgroup is the limit (beyond ~10 items per thread the stepwise form, which spreads a step over the chip, wins).
     * What fits a workgroup's LDS is kept there for the whole episode (static tables, per-step table rows, ghosts, the step's signals,
     * routes; the state rows and cotangent planes; the vehicle slots [n_micro][lane_capacity] and their cotangents): size
     * hyb.lane_capacity to the lanes (longest micro lane in vehicles + 2, dhts/stepwise.py default_lane_capacity) rather than to 32.
     * cell_lane is read by BOTH forms (the queue loss runs a thread per cell).
     * n_replicas > 1 is allowed then: action [R][A], hist [R][T + 1][4][C], queue [R][T][L], reward [R][2], counts [R][4],
     * g_reward [R], g_action [R][A]; per-replica [T][L] tables through hyb.net.replica_stride / hyb.draws_stride as in
     * dhts_net_hybrid_rollout_fwd.  cell_lane [n_cells] (device): the lane of every cell (group-major order), required whenever
     * n_cells > 0; if_lane [n_cells + ARZ lanes] (device): the lane of interface item lane_off[l] + lane_gpos[l] + k, k = 0 .. n --
     * not read by the kernels as they stand (a cell's thread solves both its interfaces), may be NULL. */
    const int32_t *if_lane, *cell_lane;
    int32_t persistent;
    int32_t n_inter_slots;             /* length of inter_idx (the persistent kernels stage the static tables in LDS) */
} dhts_netstep_tables;
size_t dhts_netstep_workspace_bytes(const dhts_net_desc *d, const dhts_netstep_tables *t);
int dhts_netstep_rollout_fwd(const dhts_net_desc *d, const dhts_netstep_tables *t, int hard, const float *action, float *hist,
                             float *queue, float *reward, int32_t *counts, void *workspace, dhts_error *err, void *stream);
int dhts_netstep_rollout_bwd(const dhts_net_desc *d, const dhts_netstep_tables *t, const float *action, const float *hist,
                             const float *queue, const float *g_reward, float *g_action, void *workspace, dhts_error *err,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DHTS_H */
