/*
 * admpc_quad.h -- C ABI of the second vehicle model behind the engine: the quadrotor of
 * data_driven_mpc/ros_gp_mpc/src/quad_mpc (nx = 13, nu = 4), one acados SQP-RTI step per call for a batch of instances.
 *
 *   reference call site (data_driven_mpc/ros_gp_mpc/src/quad_mpc/...)              replaced by
 *   ------------------------------------------------------------------------------  --------------------------
 *   quad_3d_optimizer.py:341-393   quad_dynamics (p, q, v, w dynamics)              the model inside admpc_quad_solve_batch
 *   quad_3d_optimizer.py:150-207   AcadosOcp: LINEAR_LS cost, input box, ERK, GN,    AdmpcQuadConfig
 *                                  FULL_CONDENSING_HPIPM, SQP_RTI
 *   acados_models/my_quad_acados_ocp.json  dims / W / bounds / tf                   admpc_quad_default_config
 *   quad_3d_optimizer.py:530-566   set x0 / p / solve() / get x, u                  admpc_quad_solve_batch, _ex (gp_regression_state)
 *   quad_3d_optimizer.py:207, 446-452, 485-491  one solver per GP cluster, select_gp  admpc_quad_select_cluster_batch, admpc_quad_solve_batch_routed
 *   quad_3d_optimizer.py:364-381   linear drag (rdrv_d_mat)                          cfg.rdrv
 *
 * State  x = [p(3), q_wxyz(4), v(3), w(3)],  input u = activations of the four rotors in [0, 1].
 * Conventions as in admpc.h: device pointers owned by the caller, instance-major and dense, fp64; `stream` is a hipStream_t passed
 * as void*; 0 or a negative ADMPC_E* code is returned (admpc_last_error() of admpc.h gives the message); one solve in flight per
 * handle.  Per-instance status: 0 success (also at ipm_iter_max, acados RTI semantics), 2 SQP mode not converged within sqp_iters
 * steps (iterate valid), 4 non-finite step (iterate untouched).
 */
#ifndef ADMPC_QUAD_H
#define ADMPC_QUAD_H

#include <stdint.h>
#include "admpc.h"      /* AdmpcGp, ADMPC_* codes */

#ifdef __cplusplus
extern "C" {
#endif

#define ADMPC_QUAD_NX 13
#define ADMPC_QUAD_NU 4
#define ADMPC_QUAD_NY 17
#define ADMPC_QUAD_MAX_N 24      /* N * nu <= 64 (N <= 16): one lane per input of the condensed QP, one wave per instance; up to 24 (96
                                  * inputs; the reference class defaults to n_nodes = 20): one thread per input, two waves per instance */
#define ADMPC_QUAD_GP_MAX 3      /* residual GPs: one per body-frame acceleration component */
/* Centring safeguard of the box-QP interior point (cf. ADMPC_IPM_BLOCKED_STEP of admpc.h): after a step shorter than this the next
 * iteration is a pure centring step.  0.3 for this problem: with 0.05, 1 - 2 of 4096 aggressive scenarios fall into a limit cycle of
 * Mehrotra's heuristic (mu oscillating between 2e-4 and 8e-4 with period 12 until iter_max, 0.09 off the minimiser); with 0.3 none
 * does over five scenario seeds and the mean iteration count is unchanged (9.35). */
#define ADMPC_QUAD_IPM_BLOCKED_STEP 0.3
/* Fallback mode (cf. cfg.ipm_fallback_iter of admpc.h): an instance still iterating after this many iterations -- 1 in 20 000 aggressive
 * scenarios cycles until iter_max even with the safeguard above -- starts over from the cold start and finishes WITHOUT the second-order
 * term of the corrector, on a budget of ipm_iter_max further iterations (`iters` can exceed ipm_iter_max).  Healthy instances need <= 27. */
#define ADMPC_QUAD_IPM_FALLBACK_ITER 30

typedef struct AdmpcQuadConfig {
    int32_t N;                    /* shooting intervals (reference: 10)                                            */
    int32_t ipm_iter_max;         /* qp_solver_iter_max (50)                                                       */
    double  Ts;                   /* tf / N (0.1 s)                                                                */
    double  W[ADMPC_QUAD_NY];     /* diag of the stage weight on y = [x; u] (acados scales it by Ts)               */
    double  We[ADMPC_QUAD_NX];    /* diag of the terminal weight (0 in the shipped configuration)                  */
    double  lbu[ADMPC_QUAD_NU], ubu[ADMPC_QUAD_NU];       /* hard input box (0, 1)                                 */
    double  mass;                 /* quad_3d.py:57  1.0 kg                                                         */
    double  J[3];                 /* quad_3d.py:56  diag inertia (.03, .03, .06)                                   */
    double  max_thrust;           /* quad_3d.py:40  20 N per rotor                                                 */
    double  x_f[4], y_f[4], z_l_tau[4];                   /* rotor arms and yaw-torque coefficients (:62-74)       */
    double  g;                    /* 9.81                                                                          */
    double  rdrv[3];              /* diagonal of Faessler's linear rotor-drag matrix D (quad_3d_optimizer.py:364-381, rdrv_d_mat of the
                                   * class): v' += R(q) D R(q)' v.  Zeros (the shipped generated code, default): no drag term.     */
    double  ipm_mu0, ipm_thr0, ipm_tol_comp, ipm_tol_res; /* interior point: start and stop levels (defaults 1e-8 / 1e-8: HPIPM
                                                           * mode BALANCE, what the reference runs; 1e-10 / 1e-9 for the exact minimiser) */
    double  sqp_tol;              /* solver_type "SQP" (create_ros_gp_mpc.py:63-68 for point references, quad_3d_optimizer.py:203): with
                                   * sqp_iters > 1, acados' stopping test in front of every QP but the first -- the inf-norms of the NLP's KKT
                                   * residuals (stationarity, shooting defects, input-box violation, complementarity) all <= sqp_tol
                                   * (my_quad_acados_ocp.json:2077-2080: 1e-6): status 0; not met after sqp_iters QPs: status 2
                                   * (ACADOS_MAXITER; the iterate is valid).  0: sqp_iters plain RTI steps.                       */
    /* GP residual of the acceleration (quad_3d_optimizer.py:289-327): the features are taken from z = [x with the velocity in the
     * BODY frame; u] (feat[] indexes these 17 entries; 7..16 are offered: body-frame velocity, body rates, inputs), gp[g].out in {7, 8, 9} names the body-frame acceleration component the mean
     * is added to; the sum is rotated back to the world frame:  v' += R(q) mu(z).  n_gp = 0: nominal model (the shipped code). */
    int32_t n_gp;
    int32_t sqp_iters;            /* <= 1: SQP_RTI, one step per call (the shipped setting); > 1: that many SQP steps (nlp_solver_max_iter 100) */
    AdmpcGp gp[ADMPC_QUAD_GP_MAX];
} AdmpcQuadConfig;

typedef struct AdmpcQuadSolver AdmpcQuadSolver;

/* The shipped problem: my_quad_acados_ocp.json (N = 10, tf = 1 s, W = diag(10,10,10, 0,.1,.1,.1, .05 x 6, .1 x 4), W_e = 0,
 * 0 <= u <= 1) and the vehicle of quad_3d.py ('x' configuration). */
void admpc_quad_default_config(AdmpcQuadConfig* cfg);

int admpc_quad_create(const AdmpcQuadConfig* cfg, int device, AdmpcQuadSolver** out);
void admpc_quad_destroy(AdmpcQuadSolver* s);

/* One SQP-RTI step for B instances.
 *   x0 [B][13]   yref [B][N][17] (state and input references)   yref_e [B][13]
 *   xbar [B][N+1][13], ubar [B][N][4]  in: linearisation point, out: the iterate after the full step
 *   cost [B], status [B], iters [B]  (each may be NULL) */
int admpc_quad_solve_batch(AdmpcQuadSolver* s, int B, const double* x0, const double* yref, const double* yref_e,
                           double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream);

/* The same with the GP state of the first optimisation node given per instance: gp_state [B][13] is run_optimization's
 * gp_regression_state (quad_3d_optimizer.py:546-552: parameter p = [gp_state, 1] at node 0, zeros elsewhere; :291-297: there the GP
 * features and the rotation of the GP means come from the parameter instead of the integrated state).  NULL: the initial state x0, the
 * reference's default -- which is also what admpc_quad_solve_batch does.  Ignored by a model without GPs. */
int admpc_quad_solve_batch_ex(AdmpcQuadSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* gp_state,
                              double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream);

/* Clustered GP ensembles.  The reference keeps one acados solver per cluster (quad_3d_optimizer.py:207) and picks one per solve from the
 * reference state (set_reference_state :446-452 / set_reference_trajectory :485-491 -> gp.py:738-770 select_gp: nearest centroid).
 * admpc_quad_select_cluster_batch: route[b] = index of the centroid (centroids [K][n_feat]) nearest to the selected features (feats[]
 * index z = [x with the velocity in the body frame; u], 17 entries) of (x_sel [B][13] world-frame velocity, u_sel [B][4]); ties to the
 * lower index.  admpc_quad_solve_batch_routed: one call for the batch, solvers[route[b]] solves instance b in place (every handle
 * runs over the batch and leaves the others' instances alone: no gather / scatter); an out-of-range route gives status 4, infinite
 * cost and an untouched iterate.  The solvers must share device and horizon. */
int admpc_quad_select_cluster_batch(int device, int B, int n_feat, const int32_t* feats, const double* x_sel, const double* u_sel,
                                    int K, const double* centroids, int32_t* route, void* stream);
int admpc_quad_solve_batch_routed(AdmpcQuadSolver* const* solvers, int K, int B, const int32_t* route,
                                  const double* x0, const double* yref, const double* yref_e, const double* gp_state,
                                  double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream);

/* Test hook: the shooting step alone -- phi [B][N][13], A [B][N][13][13], Bm [B][N][13][4] of every interval. */
int admpc_quad_shoot_batch(AdmpcQuadSolver* s, int B, const double* xbar, const double* ubar, double* phi, double* A, double* Bm, void* stream);
/* ... with the first node's GP state (NULL: xbar_0 of the instance) */
int admpc_quad_shoot_batch_ex(AdmpcQuadSolver* s, int B, const double* xbar, const double* ubar, const double* gp_state,
                              double* phi, double* A, double* Bm, void* stream);

#ifdef __cplusplus
}
#endif
#endif
