/*
 * dhts_oracle.h -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call this.
 * The product (diff-hybrid-traffic-sim_amd/) never includes or links anything under oracle/.
 *
 * Scalar plain-C restatement of SonSang/diff-hybrid-traffic-sim's per-step forward + adjoint of
 *   - the ARZ cell stencil  (model/macro/_arz.py, model/macro/darz.py, road/lane/_macro_lane.py,
 *                            road/lane/dmacro_lane.py)
 *   - the IDM car-following ODE (model/micro/_idm.py, model/micro/didm.py, road/lane/_micro_lane.py,
 *                            road/lane/dmicro_lane.py)
 * following the reference's precision ladder (Note P): float32 state widened to
 * double, step and Jacobian entries in double with libm pow(), float32 rounding on store, float32 2x2
 * products and adjoint.  Parity pinned against the .npz files in tests/golden, which tools/gen_goldens.py produced
 * by importing the reference in the build container (tests/test_oracle_golden.py): known answers bit for bit, rollouts and
 * networks within 1e-5 / 1e-4 at the defaults (measured <= 2.9e-6), and -- with the two library behaviours under the reference
 * switched in (oracle_set_sqrtf_hook: that torch build's float32 sqrt; oracle_set_numpy_mean: numpy's float32 summation tree) --
 * the straight-lane states of every rollout fixture and every queue term of the macro, `micro`-mode and hybrid network fixtures BIT FOR BIT.
 */
#ifndef DHTS_ORACLE_H
#define DHTS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* error codes of the step functions (the reference asserts instead: _macro_lane.py:141-146,
 * _micro_lane.py:162) */
#define ORACLE_OK 0
#define ORACLE_ERR_CFL 1
#define ORACLE_ERR_COLLISION 2
#define ORACLE_ERR_ROUTE 3        /* hybrid network: spawn without a matching pre-drawn route / lane over capacity */
#define ORACLE_E_INVALID 4        /* network episodes: fewer actions than intersections, no frames per phase (the library: DHTS_E_INVALID) */

/* ---- ARZ: one interface -------------------------------------------------------------------- */
/* L, R: (r, y, u, u_eq) of the left / right cell as doubles holding float32 values.
 * out: case_ind (0 = Q_L, 1 = Q_M, 2 = Q_C), q0 = (r, y, u, u_eq) of Q_0, speed = (speed0, speed1).
 * model/macro/_arz.py:212-332 */
void oracle_arz_riemann(const double L[4], const double R[4], double u_max,
                        int *case_ind, double q0[4], double speed[2]);
/* dQ_0/dQ_L and dQ_0/dQ_R, row-major 2x2 float32.  model/macro/darz.py:12-215 */
void oracle_arz_dLdR(int case_ind, const double q0[4], const double L[4], const double R[4], double u_max,
                     float dL[4], float dR[4]);
/* flux Jacobian at Q_0, row-major 2x2 float32.  model/macro/darz.py:217-233 */
void oracle_arz_flux_prime(const double q0[4], double u_max, float fp[4]);

/* float32 glue (torch 0-dim tensor arithmetic in the reference): model/macro/_arz.py:82-92,121-138 */
void oracle_arz_from_r_u(float r, float u, float u_max, float *y, float *u_eq);     /* FullQ.from_r_u / set_r_u */
void oracle_arz_from_r_y(float r, float y, float u_max, float *u, float *u_eq);     /* FullQ.set_r_y         */

/* ---- ARZ: one lane step (dMacroForwardLayer.forward, road/lane/dmacro_lane.py:236-275) ------- */
/* state: 4 arrays of N+2 float32 (r, y, u, u_eq), ghosts at index 0 and N+1 (pointer-per-array).
 * out: nr, ny, nu, nueq [N]; dqs [N][3][2][2]; optional case_out [N+1], speed_out [N+1][2] (NULL ok).
 * returns ORACLE_OK or ORACLE_ERR_CFL (first offending interface in *err_index if non-NULL). */
int oracle_macro_step(int N, const float *r, const float *y, const float *u, const float *ueq,
                      double dt, double dx, double u_max,
                      float *nr, float *ny, float *nu, float *nueq, float *dqs,
                      int *case_out, double *speed_out, int *err_index);
/* dMacroForwardLayer.backward (road/lane/dmacro_lane.py:277-309): g_nr, g_ny [N] -> g_r, g_y [N+2] */
void oracle_macro_step_bwd(int N, const float *dqs, const float *g_nr, const float *g_ny,
                           float *g_r, float *g_y);

/* ---- ARZ: batched straight-lane rollout (what example/inverse/macro.py + RoadNetwork.forward do
 *      for one dMacroLane with fixed ghosts) -------------------------------------- */
/* L independent lanes of N cells.  r0, u0 [L][N]; ghost_r, ghost_u [L][2] (left, right).
 * Forward: T steps; outputs rT, yT, uT [L][N]; tape [T][L][N][12] (caller-allocated) or NULL when no
 * gradient is wanted; hist_r/hist_y/hist_u [T][L][N] optional (NULL ok) = state after each step.
 * returns ORACLE_OK / ORACLE_ERR_CFL. */
int oracle_macro_rollout_fwd(int L, int N, int T, double dt, double dx, double u_max,
                             const float *r0, const float *u0, const float *ghost_r, const float *ghost_u,
                             float *rT, float *yT, float *uT, float *tape,
                             float *hist_r, float *hist_y, float *hist_u);
/* Reverse sweep.  Cotangents on the final state g_rT, g_yT, g_uT [L][N] (any may be NULL = zero);
 * optional per-step cotangents gh_r, gh_y, gh_u [T][L][N] on the state after each step (need hist_r,
 * hist_y from the forward when gh_u is given).  Outputs g_r0, g_u0 [L][N], g_ghost_r, g_ghost_u [L][2]. */
void oracle_macro_rollout_bwd(int L, int N, int T, double u_max,
                              const float *tape, const float *r0, const float *u0,
                              const float *ghost_r, const float *ghost_u,
                              const float *rT, const float *yT,
                              const float *g_rT, const float *g_yT, const float *g_uT,
                              const float *hist_r, const float *hist_y,
                              const float *gh_r, const float *gh_y, const float *gh_u,
                              float *g_r0, float *g_u0, float *g_ghost_r, float *g_ghost_u);

/* ---- IDM -------------------------------------------------------------------------------------- */
/* model/micro/_idm.py:6-50.  returns acc; *sstar = clipped optimal spacing; flags[0] = clipped_acc,
 * flags[1] = clipped_spacing */
double oracle_idm_acc(double a_max, double a_pref, double v, double v_target, double dp, double dv,
                      double min_space, double time_pref, double dt, double *sstar, int flags[2]);
/* model/micro/didm.py:13-103: row-major 2x2 float32 each */
void oracle_idm_jac(double a_max, double a_pref, double v, double v_target, double dp, double dv,
                    double min_space, double time_pref, double sstar, double dt, const int flags[2],
                    float dEgo[4], float dLeading[4]);

/* one lane step (dMicroForwardLayer.forward, road/lane/dmicro_lane.py:230-269).
 * p, v [V] float32 (index i follows i+1, head = V-1); params [V][6] double =
 * (a_max, a_pref, v_target, min_space, time_pref, length); head gap (head_dp, head_dv).
 * out np_, nv_ [V]; dqs [V][2][2][2].  returns ORACLE_OK / ORACLE_ERR_COLLISION. */
int oracle_micro_step(int V, const float *p, const float *v, const double *params,
                      double head_dp, double head_dv, double dt,
                      float *np_, float *nv_, float *dqs, int *err_index);
/* The same step as the reference's PLAIN MicroLane computes it on torch tensors (itscp `micro` mode, differentiable episodes): every
 * operation in float32 (road/lane/_micro_lane.py:131-214 + model/micro/_idm.py:30-50 evaluated by torch). */
int oracle_micro_step_f32(int V, const float *p, const float *v, const double *params,
                          double head_dp, double head_dv, double dt,
                          float *np_, float *nv_, float *dqs, int *err_index);
/* dMicroForwardLayer.backward (dmicro_lane.py:271-298): g_np, g_nv [V] -> g_p, g_v [V+1] */
void oracle_micro_step_bwd(int V, const float *dqs, const float *g_np, const float *g_nv,
                           float *g_p, float *g_v);

/* batched rollout: L lanes of V vehicles each, fixed head gap.  p0, v0 [L][V]; params [L][V][6];
 * tape [T][L][V][8] or NULL; hist_p/hist_v [T][L][V] optional. */
int oracle_micro_rollout_fwd(int L, int V, int T, double dt, const float *p0, const float *v0,
                             const double *params, double head_dp, double head_dv,
                             float *pT, float *vT, float *tape, float *hist_p, float *hist_v);
/* g_pT, g_vT [L][V]; optional per-step gh_p, gh_v [T][L][V]; out g_p0, g_v0 [L][V] and the cotangent
 * wrt (head_position_delta, head_speed_delta) g_head [L][2], summed over steps (NULL ok). */
void oracle_micro_rollout_bwd(int L, int V, int T, const float *tape,
                              const float *g_pT, const float *g_vT, const float *gh_p, const float *gh_v,
                              float *g_p0, float *g_v0, float *g_head);


/* ---- macro road NETWORK with differentiable signals (itscp `macro` mode) ----------------------------------------
 * Restates what ItscpEnv._simulate + ItscpRoadNetwork.forward do for a network of dMacroLanes
 * (example/control/itscp/_env.py:620-768, 885-962; _simulator.py:56-137; road/network/road_network.py:79-111,299-387):
 * per step  signals from the action -> ghost cells of every lane from the time-n state of its neighbours (Jacobi),
 * blended between green and red values -> one ARZ step per lane -> queue-length loss with the running-mean-scaled
 * sigmoid (_env.py:586-618, example/common/rms.py).
 * Topology and schedules come as tables (built by the caller from the environment):
 *   lane_ncell, lane_off [L] (cells and first-cell offset), lane_dx [L] double, sig_kind [L] (0 = always green,
 *   1 = west-east phase, 2 = north-south phase), inter [L] (intersection index), per step t and lane l:
 *   left_src  (lane whose LAST cell is the green upstream ghost; -1 = source lane: r = schedule, u = u_eq(r)),
 *   left_gate (lane whose signal gates the upstream ghost; -1 = red (0.0); -2 = always 1.0),
 *   right_src (lane whose FIRST cell is the green downstream ghost; -1 = the lane's own stored ghost),
 *   schedule [T][L] double (inflow density of source lanes).
 * All lanes start empty (r = 0, y = 0, u = u_eq = u_max).  Work arrays are caller-allocated:
 *   hist [T+1][4][C] float (r, y, u, u_eq after t steps), tape [T][C][12], kc [T][C] float (sigmoid constants),
 *   queue [T][L] float (per-lane loss terms q^2 dt).  Returns reward = - sum queue in *reward (float32 accumulation
 *   order of the reference: lanes outer, steps inner).  rc = ORACLE_OK / ORACLE_ERR_CFL. */
/* 1 = the network forward passes below run EVALUATION episodes (hard thresholds, ItscpEnv.step(action, False)); their
 * gradient outputs are then meaningless.  Process-wide switch of this test library. */
void oracle_set_hard(int hard);
/* micro SOURCE lanes of oracle_net_hybrid (itscp `micro` mode, _simulator.py:153-174): lane_source [L] (1 = micro lane without an
 * upstream lane), the host's admission draws in call order; NULL = none.  Process-wide, set before the call. */
void oracle_set_micro_sources(const int *lane_source, const double *draws, int n_draws);
void oracle_set_micro_tensor_ladder(int on);
void oracle_set_vehicle_params(const double *params);      /* [n_routes][6] beside the route table, or NULL */
int oracle_micro_source_draws_used(void);
typedef struct oracle_net_desc {
    int n_lanes, n_cells, T, n_inter_sq, frames_per_phase, n_action;
    double dt, u_max, static_speed, vehicle_length;
} oracle_net_desc;
int oracle_net_macro_fwd(const oracle_net_desc *d, const int *lane_ncell, const int *lane_off, const double *lane_dx,
                         const int *sig_kind, const int *inter, const int *left_src, const int *left_gate,
                         const int *right_src, const double *schedule, const float *action,
                         float *hist, float *tape, float *kc, float *queue, double *reward);
/* d reward / d action [n_action] */
void oracle_net_macro_bwd(const oracle_net_desc *d, const int *lane_ncell, const int *lane_off, const double *lane_dx,
                          const int *sig_kind, const int *inter, const int *left_src, const int *left_gate,
                          const int *right_src, const double *schedule, const float *action,
                          const float *hist, const float *tape, const float *kc, const float *queue, float *g_action);

/* ---- HYBRID road network (itscp `hybrid` mode): macro lanes + micro lanes + hand-offs ------------------------------
 * As oracle_net_macro_* plus, per lane: lane_macro [L] (1 = ARZ cells, 0 = IDM vehicles; micro lanes have 0 cells),
 * lane_len [L] double, and
 *   left_src   additionally -3 = the lane's own stored upstream ghost (its single upstream lane is micro; with
 *              left_gate -1 the blended ghost is the red value (0, u_max)),
 *   right_src  -1 also for a macro lane whose single downstream lane is micro (own stored ghost, _simulator.py:116),
 *   conv_next  [T][L] the step's macro-route successor of a macro lane (conversion target, road_network.py:113-130),
 *   routes     [n_routes][route_stride] lane ids (-1 padded) grouped by first lane, route_ptr [L+1]: the k-th vehicle
 *              spawned onto lane m takes row route_ptr[m] + k mod (rows of m) (the reference draws them with np.random
 *              at spawn time, road_network.py:604-646; the caller pre-draws / replays them).  At most 128 vehicles per episode.
 * Forward and reverse sweep in one call: queue [T][L], *reward (all steps), *reward_cut and g_action
 * (d reward_cut / d action, NULL = forward only) for the reward restricted to steps < t_cut; *n_spawned, *n_deposits;
 * hist_out [T+1][4][C] and kc_out [T][C] (the cells' loss sigmoid constants) optional.  rc = ORACLE_OK / ORACLE_ERR_CFL / ORACLE_ERR_ROUTE. */
int oracle_net_hybrid(const oracle_net_desc *d, const int *lane_macro, const double *lane_len,
                      const int *lane_ncell, const int *lane_off, const double *lane_dx,
                      const int *sig_kind, const int *inter, const int *left_src, const int *left_gate,
                      const int *right_src, const int *conv_next, const double *schedule,
                      const int *routes, const int *route_ptr, int n_routes, int route_stride, const float *action, int t_cut,
                      float *queue, double *reward, double *reward_cut, float *g_action, int *n_spawned,
                      int *n_deposits, float *hist_out, float *kc_out);

#ifdef __cplusplus
}
#endif
/* 1 = the itscp network oracles evaluate their running means as numpy does on the reference's float32 array (pairwise float32
 * summation, O(window) per sample); 0 (default) = exact float64 prefix means */
void oracle_set_numpy_mean(int on);
/* 1 (default) = the upstream ghost of an itscp source lane enters the solve in double, as the reference's Python floats do
 * (_simulator.py:68-71) -- in oracle_net_macro_fwd and oracle_net_hybrid alike; 0 = its float32 rounding (oracle and kernels until the
 * end of round 5) */
void oracle_set_source_ghost_f64(int on);
/* the float32 square root of the glue (u_eq of a float32 tensor) as the caller's environment evaluates it; NULL = sqrtf (dhts_oracle.c) */
void oracle_set_sqrtf_hook(float (*f)(float));
/* debugging aid of tools/probes/ref_state_trace.py: oracle_net_hybrid writes (lane or -1, position, speed) of vehicle ids < n_vehicles after
 * every step into buf [T][n_vehicles][3]; NULL = off */
void oracle_set_vehicle_trace(float *buf, int n_vehicles);
/* the head vehicle of an IDM lane of a differentiable itscp hybrid episode: mixed float32 / double arithmetic (dhts_oracle.c); the
 * switch (default 1) is read by oracle_net_hybrid */
void oracle_micro_head_mixed(int V, const float *p, const float *v, const double *params, float head_dp, float head_dv, double dt,
                             float *np_, float *nv_, float *dqs);
void oracle_set_head_mixed(int on);

#endif
