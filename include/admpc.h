/*
 * admpc.h -- C ABI of the MI355X-native batched AD-MPC solve engine (libadmpc.so).
 *
 * This is the drop-in boundary for the ros_gp_mpc AD-MPC inner loop.  It replaces, for a
 * whole batch of independent MPC instances at once, what the reference reaches through the
 * acados Python/ctypes seam:
 *
 *   reference call site (data_driven_mpc/ros_gp_mpc/src/ad_mpc/...)        replaced by
 *   ---------------------------------------------------------------------  ------------------------
 *   ad_3d_optimizer.py:135-209  AcadosOcp setup + AcadosOcpSolver(ocp)      AdmpcConfig + admpc_create
 *   c_generated_code/acados_solver_sim_car.c:343-699  (dims/cost/bounds/opts)  AdmpcConfig fields
 *   ad_3d_optimizer.py:420-450  solver.set(j,"yref"/"lbx"/"ubx"/"p")       device arrays yref/yref_e/x0/p
 *   ad_3d_optimizer.py:456      solver.solve()  (acados SQP_RTI step)       admpc_solve_batch
 *   ad_3d_optimizer.py:460-465  solver.get(i,"u"/"x")                       in-out arrays xbar/ubar
 *   c_generated_code/acados_solver_sim_car.h:115-146  sim_car_acados_*      admpc_create/_destroy/_solve_batch
 *   (new capability, BASELINE.json north_star)  arg-min over scenario cost  admpc_argmin
 *
 * Conventions
 *   - plain C, no torch / HIP types in the signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).
 *   - all array arguments are DEVICE pointers owned by the caller, instance-major and densely
 *     packed; the library never allocates per call, never synchronises the stream, never exits
 *     the process.
 *   - every function returns 0 on success or a negative ADMPC_E* code; admpc_last_error()
 *     gives a thread-local human readable message.
 *   - per-instance solver status follows the acados enum the reference caller tests
 *     (gp_ad_mpc_node.py:206-209 treats status>0 as failure): 0 success, 4 QP failure / NaN.
 *     As in acados SQP_RTI, a QP that stops at ipm_iter_max is still reported as 0;
 *     the iteration count is available through `iters`.
 */
#ifndef ADMPC_H
#define ADMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADMPC_NX 7          /* p_x p_y psi v_x v_y psi_dot delta   (ad_3d_optimizer.py:74-87)  */
#define ADMPC_NU 2          /* a, delta_dot                        (ad_3d_optimizer.py:90-93)  */
#define ADMPC_NY 9          /* y = [x;u]                           (ad_3d_optimizer.py:156-159)*/
#define ADMPC_MAX_N 128     /* largest supported horizon                                       */
#define ADMPC_GP_MAX 4      /* residual GPs per model                                          */
#define ADMPC_GP_MAX_POINTS 32
#define ADMPC_GP_MAX_FEAT 3   /* features per residual GP                                         */

#define ADMPC_OK            0
#define ADMPC_EINVAL      (-1)   /* bad argument / unsupported configuration  */
#define ADMPC_ENODEV      (-2)   /* no usable HIP device                      */
#define ADMPC_EHIP        (-3)   /* HIP runtime error (see admpc_last_error)  */
#define ADMPC_ENOMEM      (-4)

/* Safeguard of the Mehrotra predictor-corrector (oracle and every device path): after an iteration whose step length was below
 * this value the next iteration is a pure centring step (sigma = 1).  Without it the method can fall into a limit cycle on a
 * badly centred iterate (seen on one of 16384 config-5 scenarios: mu cycling with period 4 until iter_max). */
#define ADMPC_IPM_BLOCKED_STEP 0.05

/* Two more safeguards (round 3; oracle and every device path of the car model), both against the same failure: once the
 * complementarity products are below their tolerance while the last step was not yet below its own, the method takes another
 * iteration, and Mehrotra's heuristic (sigma = (mu_aff / mu)^3 ~ 1e-15 by then) sends mu from 1e-11 to 1e-17 .. 1e-27.  The Newton
 * matrix of that iteration has barrier ratios lam / t beyond 1e16, and the linear algebra loses every digit: on one of 36 864 GP
 * scenarios the condensed factorisation produced a garbage direction (a blocked step that moved the inputs by 1.3e-2, accepted by
 * the rounding-floor clause of the stopping test); on a long-horizon fallback instance the Riccati form's residual grew from 1e-9
 * to 1e10 until the iteration limit.
 *   ADMPC_IPM_MU_FLOOR   the centring target sigma * mu is never below MU_FLOOR * ipm_tol_comp: the method does not drive the
 *                        complementarity further below its tolerance than three orders of magnitude;
 *   ADMPC_IPM_FLOOR_CAP  stopping test:  max lam*t <= ipm_tol_comp  and  last input step <= ipm_tol_step  and
 *                        (residual <= ipm_tol_res  or  the residual sits on its rounding floor -- no longer falling by 10x per
 *                        iteration -- AND is not above FLOOR_CAP * ipm_tol_res): the floor clause never accepts a large residual. */
#define ADMPC_IPM_MU_FLOOR 1e-3
#define ADMPC_IPM_FLOOR_CAP 1e3

/* status per instance (acados enum values used by the reference) */
#define ADMPC_STATUS_SUCCESS     0
#define ADMPC_STATUS_MAXITER     2   /* SQP mode with sqp_tol > 0: not converged within sqp_iters steps (iterate and cost are valid) */
#define ADMPC_STATUS_QP_FAILURE  4

/* One squared-exponential residual GP with an anisotropic (diagonal) length scale over 1..3 features:
 * f[out] += mu(z),  z_d = [x;u][feat[d]]
 *   mu(z) = ymean + sum_i sigma_f * exp(-0.5 * sum_d (z_d - Z[d][i])^2 * inv_l2[d]) * alpha_i
 * (model_fitting/gp.py:81-138,446-471 -- note sigma_f is NOT squared, gp.py:138;
 *  B_z/B_x selection matrices utils/utils.py:773-808; wiring quad_3d_optimizer.py:289-327) */
typedef struct AdmpcGp {
    int32_t n_feat;                     /* 1 .. ADMPC_GP_MAX_FEAT (x_features + u_features of the saved regressor, gp.py:495-508) */
    int32_t feat[ADMPC_GP_MAX_FEAT];    /* indices into [x(7);u(2)], each in {3..8}: v_x v_y psi_dot delta a delta_dot */
    int32_t out;                        /* state derivative row that receives mu */
    int32_t n_points;                   /* <= ADMPC_GP_MAX_POINTS */
    double  sigma_f;
    double  inv_l2[ADMPC_GP_MAX_FEAT];  /* 1 / l_d^2 (a scalar length scale is repeated) */
    double  ymean;
    double  Z[ADMPC_GP_MAX_FEAT][ADMPC_GP_MAX_POINTS];   /* training inputs, feature-major */
    double  alpha[ADMPC_GP_MAX_POINTS]; /* K^-1 y */
} AdmpcGp;

/* Problem description; mirrors RGM/acados_models/sim_car_acados_ocp.json and
 * acados_solver_sim_car.c one to one. */
typedef struct AdmpcConfig {
    int32_t N;                 /* horizon, ocp.dims.N                       (ad_3d_optimizer.py:139) */
    int32_t ipm_iter_max;      /* qp_iter_max = 50                (acados_solver_sim_car.c:692)      */
    int32_t sqp_iters;         /* 1 = SQP_RTI (ad_3d_optimizer.py:205); >1 = that many full SQP steps*/
    int32_t n_gp;              /* 0 = nominal model                                                  */
    double  Ts;                /* tf/N, also the stage cost scaling (acados_solver_sim_car.c:362-366)*/
    double  W[ADMPC_NY];       /* diag of ocp.cost.W  = [q;r]              (ad_3d_optimizer.py:149)  */
    double  We[ADMPC_NX];      /* diag of ocp.cost.W_e = q*1e-6            (ad_3d_optimizer.py:151)  */
    double  lbu[ADMPC_NU];     /* [acc_min, steering_rate_min]             (ad_3d_optimizer.py:184)  */
    double  ubu[ADMPC_NU];     /* [acc_max, steering_rate_max]             (ad_3d_optimizer.py:185)  */
    double  lbx_delta;         /* steering_min, state index 6, stages 1..N-1 (ad_3d_optimizer.py:188-190)*/
    double  ubx_delta;
    double  zl;                /* L1 slack penalty on u bounds, =10; multiplied by Ts (py:171-174)   */
    double  zu;
    /* vehicle (ad_3d.py:47-64) */
    double  mass, L_F, L_R, Iz, Cf, Cr;
    /* interior point options (our own; the reference delegates to HPIPM "BALANCE") */
    double  ipm_mu0;           /* initial complementarity target                                     */
    double  ipm_thr0;          /* lower clip of the initial slacks                                   */
    double  ipm_tol_comp;      /* stop when max_i lam_i*t_i <= tol_comp ...                          */
    double  ipm_tol_res;       /* ... and max |linear KKT residual| <= tol_res (or it sits on its     */
                               /*     rounding floor) ...                                            */
    double  ipm_tol_step;      /* ... and the last applied input step max|alpha*ddu| <= tol_step      */
                               /* Defaults: the reference's levels -- it leaves qp_solver_tol_* unset and runs HPIPM in mode BALANCE
                                * (acados_solver_sim_car.c:688): 1e-8 on every residual norm and on the complementarity products, no step
                                * test: tol_comp = tol_res = 1e-8, tol_step = 1e30 (off).  Tighter levels (1e-10 / 1e-9 / 1e-6, the
                                * defaults of earlier versions) take instances with a nearly degenerate bound pair from ~1e-4 to within
                                * 1e-8 of the exact minimiser for one to four more iterations.  fp32 entry point: floors 1e-3 / 1e-2 and
                                * ALWAYS a step test (the configured one when it lies in [1e-3, 1], else 1e-3), applied to the Newton step
                                * max|ddu| itself: a blocked step (alpha ~ 1e-3) is short without being converged. */
    double  ipm_try_unconstrained; /* != 0: first solve the QP without its inequalities (one factorisation + one solve);
                                    * if that minimiser respects every bound it IS the QP solution (iters = 0) and the
                                    * interior point is skipped.  Default 1 (all device paths and the oracle). */
    double  ipm_warm_thr;      /* > 0 (and the trial on): when that minimiser violates a bound the interior point starts
                                * from it instead of from the zero step -- inputs, states and dynamics multipliers of the
                                * minimiser, input slacks sl/su = violation + ipm_warm_thr, every inequality slack clipped
                                * below at ipm_warm_thr, multipliers ipm_mu0 / slack.  0: cold start.  Default 0.01. */
    double  ipm_warm_restart;  /* a warm start whose FIRST interior-point step is shorter than this is abandoned: the instance starts
                                * over from the cold start and the iteration counts.  From a minimiser far outside the hard steering
                                * box the method otherwise creeps for about ten blocked steps (N = 80: the slowest instances take
                                * 24 iterations warm, 17 cold; no instance of the N = 20 / 40 batches is affected).  Default 0.1;
                                * 0: never. */
    double  ipm_fallback_iter; /* an instance that is still iterating after this many interior-point iterations has most likely fallen
                                * into a limit cycle of Mehrotra's heuristic (seen on about 1 in 20 000 long-horizon scenarios: mu
                                * cycling with period 4 around 1e-4 behind one badly centred pair, until iter_max).  It starts over from
                                * the cold start and finishes WITHOUT the second-order corrector term (plain predictor-centring steps,
                                * which do not cycle); the iterations keep counting and the instance may use ipm_iter_max further iterations.
                                * Default 30 (healthy instances need up to 18 / 23 / 27 iterations at N = 40 / 80 / 128:
                                * scripts/sweep_convergence.py); 0: never. */
    double  sqp_tol;           /* sqp_iters > 1 only (reference solver_type "SQP", create_ros_ad_mpc.py:47-51): > 0 is acados' stopping test with
                                * all four of nlp_solver_tol_{stat,eq,ineq,comp} set to this value (the reference leaves them at acados'
                                * default 1e-6: acados_models/sim_car_acados_ocp.json:870-873).  In front of every QP of a solve but the first
                                * the iterate is linearised and the inf-norms of the NLP's KKT residuals are formed with the multipliers of
                                * the previous QP (see admpc_nlp_residuals_batch); an instance whose four norms are within sqp_tol stops with
                                * status 0, one that has not got there after sqp_iters QPs returns status 2 (acados ACADOS_MAXITER) with
                                * its last iterate.  Such solves run on the row kernel at every horizon (the condensed N = 20 kernels carry
                                * no multipliers).  fp32 entry point: the tolerances are floored at the float QP's own stop levels (1e-2
                                * stationarity / inequalities, 1e-3 complementarity, 1e-4 defects).  0: always sqp_iters steps. */
    AdmpcGp gp[ADMPC_GP_MAX];
} AdmpcConfig;

typedef struct AdmpcSolver AdmpcSolver;   /* opaque */

/* Fill `cfg` with the reference's shipped values for horizon N and sampling time Ts
 * (W, We=1e-6*q, bounds, slack penalty, vehicle constants, IPM defaults, no GP). */
int admpc_default_config(AdmpcConfig* cfg, int N, double Ts);

/* Create / destroy a solver bound to HIP device `device`.  Replaces AcadosOcpSolver(ocp)
 * (ad_3d_optimizer.py:209) / sim_car_acados_create (acados_solver_sim_car.h:119). */
int  admpc_create(const AdmpcConfig* cfg, int device, AdmpcSolver** out);
void admpc_destroy(AdmpcSolver* s);

/* Pre-allocate the internal linearisation workspace (49 doubles per stage and instance) for batches of
 * up to B instances.  admpc_solve_batch grows it on demand (which allocates and synchronises); calling
 * this once up front keeps every later solve allocation-free and stream-capture safe. */
int admpc_reserve(AdmpcSolver* s, int B);

/* One SQP real-time iteration (or cfg.sqp_iters full steps) for B independent instances.
 *   x0     [B][7]        measured state             (solver.set(0,'lbx'/'ubx',x0), py:441-442)
 *   yref   [B][N][9]     stage references [x;u]     (solver.set(j,'yref',ref),     py:430)
 *   yref_e [B][7]        terminal reference         (solver.set(N,'yref',...),     py:438)
 *   p      [B]           blend switch in [0,1]      (solver.set(j,'p',[vel_switch]), py:443-450)
 *   xbar   [B][N+1][7]   in: current iterate, out: new iterate  (solver.get(i,'x'), py:462-465)
 *   ubar   [B][N][2]     in: current iterate, out: new iterate  (solver.get(i,'u'), py:464)
 *   cost   [B]           objective of the returned iterate (+inf if status != 0)   (may be NULL)
 *   status [B]           acados-style status                                        (may be NULL)
 *   iters  [B]           interior-point iterations of the last QP                   (may be NULL)
 * Asynchronous on `stream`.
 * ONE in-flight solve per handle: the handle owns one linearisation / QP workspace and one scheduler array, which every call
 * rewrites.  Use one handle per stream; a call that has to grow the workspace (B above anything reserved so far) synchronises
 * the device first. */
int admpc_solve_batch(AdmpcSolver* s, int B,
                      const double* x0, const double* yref, const double* yref_e, const double* p,
                      double* xbar, double* ubar,
                      double* cost, int32_t* status, int32_t* iters,
                      void* stream);

/* admpc_solve_batch plus the multipliers of the returned iterate -- what acados' store_iterate writes next to x and u
 * (format of src/ad_mpc/sim_car_iterate.json; SURVEY 8c pin 2):
 *   pi   [B][N+1][7]  rows 0..N-1: multipliers pi_k of x_{k+1} = phi(x_k, u_k); row N: multiplier of the initial-state equality
 *                     (acados splits it into the lbx / ubx multipliers of stage 0: positive part -> lbx, negative part -> ubx)
 *   ineq [B][N][20]   per stage: slacks t[10] then multipliers lam[10] of the pairs
 *                     {lbu0, ubu0, lbu1, ubu1, lbx(delta), ubx(delta), sl0>=0, su0>=0, sl1>=0, su1>=0}
 *                     (stage 0 has no steering bound: slack 1, multiplier 0 there)
 * Both NULL: identical to admpc_solve_batch.  Both given: the step runs on the row kernel at every horizon. */
int admpc_solve_batch_ex(AdmpcSolver* s, int B,
                         const double* x0, const double* yref, const double* yref_e, const double* p,
                         double* xbar, double* ubar,
                         double* cost, int32_t* status, int32_t* iters,
                         double* pi, double* ineq, void* stream);

/* The four residuals of acados' SQP stopping test (what AcadosOcpSolver.get_residuals() returns: res_stat, res_eq, res_ineq, res_comp)
 * at an iterate (xbar, ubar) with the multipliers pi / ineq that admpc_solve_batch_ex returned for it:
 *   res [B][4]   inf-norms of  {gradient of the Lagrangian (states, inputs, slack variables), shooting defects and x0 - x_0,
 *                               constraint values minus their slacks t, lam .* t}
 * The library linearises at the iterate (kernel A) and evaluates the rows on the device (admpc_nlp_res_kernel, the kernel that decides
 * convergence inside an SQP solve with cfg.sqp_tol > 0).  Batches whose linearisation would pass 4 GB are refused (ADMPC_EINVAL). */
int admpc_nlp_residuals_batch(AdmpcSolver* s, int B,
                              const double* x0, const double* yref, const double* yref_e, const double* p,
                              const double* xbar, const double* ubar, const double* pi, const double* ineq,
                              double* res, void* stream);

/* The same step in fp32 storage AND arithmetic (BASELINE configs[4]: long horizons, large batches).  Arguments as above with
 * float arrays.  The interior point of this entry stops at fp32 levels (complementarity 1e-3, residual 1e-2, last step 1e-3;
 * tighter values in cfg are clipped to these) -- the result is the fp64 minimiser to about 1e-3 of the input range.
 * The model's "+1e-99" denominators (ad_3d_optimizer.py:290,296-297) vanish in fp32: with p == 0 the dynamic branch is dropped
 * instead of multiplied by 0 (it would be inf * 0 at v_x = 0); 0 < p <= 1 reproduces the blend.
 * Every horizon runs the row kernel here (the condensed N = 20 pipeline is fp64 only). */
int admpc_solve_batch_f32(AdmpcSolver* s, int B,
                          const float* x0, const float* yref, const float* yref_e, const float* p,
                          float* xbar, float* ubar,
                          float* cost, int32_t* status, int32_t* iters,
                          void* stream);

/* Shooting only (H1): phi, A, B for every stage of every instance; used by the parity tests.
 *   xbar [B][N+1][7], ubar [B][N][2], p [B]  ->  phi [B][N][7], A [B][N][7][7] row-major,
 *   Bm [B][N][7][2] row-major. */
int admpc_shoot_batch(AdmpcSolver* s, int B,
                      const double* xbar, const double* ubar, const double* p,
                      double* phi, double* A, double* Bm, void* stream);

/* Local arg-min over cost[0..B): writes the smallest cost and its index (+index_offset) into
 * the device scalars val/idx; ties -> lowest index; +inf / NaN costs never win unless all are.
 * The cross-GPU step (RCCL all-gather of the (val, idx) pairs) is done by the host mirror in
 * ad_mpc_amd/dist.py with torch.distributed. */
int admpc_argmin(AdmpcSolver* s, const double* cost, int B, int64_t index_offset,
                 double* val, int64_t* idx, void* stream);

/* Second level of the arg-min: W (cost, global index) pairs of 16 bytes each -- val and idx of admpc_argmin written next to
 * each other by every GPU, all-gathered into one array -- reduced with the same rules to the device scalars val/idx.
 * (val and idx of admpc_argmin may point into one such 16-byte pair; the index travels as the bit pattern of an int64.) */
int admpc_argmin_pairs(AdmpcSolver* s, const double* pairs, int W, double* val, int64_t* idx, void* stream);

/* Host-memory twin of admpc_argmin_pairs: the same reduction (same source: csrc/argmin_rule.h) over W records that an all-gather
 * left in HOST memory (a gloo / MPI host, the world-size-2 CPU tests of the N > 1 path).  pairs, val, idx are host pointers; no
 * device and no solver handle are touched.  It reduces 16-byte records only -- it is not a CPU path of the solve. */
int admpc_argmin_pairs_host(const double* pairs, int W, double* val, int64_t* idx);

/* The whole cross-GPU arg-min for a host that is not Python (one process per GPU): local admpc_argmin into a 16-byte
 * (cost, global index) record, ncclAllGather of the records over `nccl_comm` (an ncclComm_t of RCCL, passed as void*; 16 B per rank:
 * the only collective of the whole path), admpc_argmin_pairs over the gathered records -- three operations on `stream`, no host
 * synchronisation.  Every rank receives the winner in val / idx (device scalars).  RCCL is looked up at the first call: first among the
 * libraries the process has already loaded (the communicator belongs to the copy that created it -- torch bundles its own librccl.so.1),
 * then by dlopen of librccl.so.1 / librccl.so; ADMPC_ENODEV when it is not installed. */
int admpc_argmin_global(AdmpcSolver* s, const double* cost, int B, int64_t index_offset, void* nccl_comm,
                        double* val, int64_t* idx, void* stream);

/* Clustered GP ensembles.  The reference keeps one solver per cluster of its GP ensemble (quad_3d_optimizer.py:207) and
 * picks one per solve with GPEnsemble.select_gp (gp.py:738-770; called at quad_3d_optimizer.py:452, :491): nearest centroid of the
 * ensemble's feature vector z = [x; u][feats].  For a batch:
 *   admpc_select_cluster_batch   route[b] = index of the centroid nearest to z_b (Euclidean distance, ties -> lowest index);
 *                                feats: n_feat (1..3) host indices into [x(7); u(2)]; x_sel [B][7], u_sel [B][2], centroids [K][n_feat],
 *                                route [B] int32: device arrays;
 *   admpc_solve_batch_routed     admpc_solve_batch with one handle per cluster: instance b is solved by solvers[route[b]] (same device,
 *                                same horizon; each handle carries the GP of its cluster).  In place on the full-size arrays, no gather /
 *                                scatter and no host synchronisation: every handle runs over the batch and leaves the instances of the
 *                                other clusters alone.  An instance with route[b] outside [0, K) is reported as status 4, cost +inf.
 * Both asynchronous on `stream`. */
int admpc_select_cluster_batch(int device, int B, int n_feat, const int32_t* feats, const double* x_sel, const double* u_sel,
                               int K, const double* centroids, int32_t* route, void* stream);
int admpc_solve_batch_routed(AdmpcSolver* const* solvers, int K, int B, const int32_t* route,
                             const double* x0, const double* yref, const double* yref_e, const double* p,
                             double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream);

/* Receding-horizon shift of the iterate between two solves.  The reference never shifts its iterate
 * (the acados capsule keeps it as it is, acados_solver_sim_car.c:705-731; reset_mpc_optimizer is a stub,
 * gp_ad_mpc_node.py:154-158), so this is an option the caller turns on, never part of admpc_solve_batch.
 *   xbar [B][N+1][7], ubar [B][N][2] in place: stage k <- stage k+1; the last input is kept;
 *   new terminal state = old terminal state (rollout = 0) or one RK4 step of the model (p [B]: blend
 *   parameter, GP residual included when configured) from it under the last input (rollout = 1). */
int admpc_shift_batch(AdmpcSolver* s, int B, double* xbar, double* ubar, const double* p, int rollout, void* stream);

/* Post-solve epilogue: validity test of ad_3d_optimizer.py:385-394 and the
 * Ackermann mapping of create_ros_ad_mpc.py:95-98 for every instance.
 *   xopt [B][N+1][7], uopt [B][N][2], xref_xy [B][N+1][2]
 *   -> ack [B][4] = {steering_angle, steering_angle_velocity, speed, acceleration}, valid [B] */
int admpc_epilogue_batch(AdmpcSolver* s, int B,
                         const double* xopt, const double* uopt, const double* xref_xy,
                         float* ack, int32_t* valid, void* stream);

/* Post-solve safety and actuation, the whole branch the node runs after optimize() -- one record per vehicle slot
 * (or per candidate of a scenario batch):
 *   check_pred_trj (gp_ad_mpc_node.py:248-257)  -> valid [B]
 *   consecutive-success gate (:206-213): safe_count [B] is in/out -- status > 0 resets it, otherwise it is incremented; an MPC
 *     command is issued only with safe_count >= threshold (the node uses 10) AND a healthy prediction
 *   steering command (:222-223): steering_angle = clip(clip(u1, rate bounds) * 0.1 + steer_meas, steering bounds)
 *   otherwise the auxiliary controller's record (:455-476): steering held at steer_meas, acceleration -1e5, the rest 0
 *   -> ack [B][4] = {steering_angle, steering_angle_velocity, speed, acceleration} (float32 message fields), mode [B] 1 = MPC, 0 = brake
 *   cost_io [B] (may be NULL): set to +inf wherever mode == 0, so that admpc_argmin over it picks among valid candidates only.
 * Bounds come from cfg (lbu[1] / ubu[1] = steering-rate bounds, lbx_delta / ubx_delta = steering bounds: ad_3d.py:66-71). */
int admpc_actuation_batch(AdmpcSolver* s, int B, const double* xopt, const double* uopt, const double* xref_xy, const int32_t* status,
                          const double* steer_meas, int32_t* safe_count, int threshold, double* cost_io,
                          float* ack, int32_t* mode, int32_t* valid, void* stream);

/* Speed-reference clamp in front of the solve: gp_ad_mpc_node.py:344-349 resample_vel, in place:
 *   bound = sqrt(vx^2 + vy^2); for i < H: vel_ref[i] = min(vel_ref[i], bound); bound += acc_max * dt * 0.8
 * vel_ref: B rows of H values, consecutive rows `ld` values apart (row 3 of admpc_waypoints_batch's out_ref: ld = 6 * H). */
int admpc_resample_vel_batch(int device, int B, int H, int ld, const double* vx, const double* vy, double acc_max, double dt,
                             double* vel_ref, void* stream);

/* Local reference generator: batched RefTrajectory.get_waypoints (src/ad_mpc/ref_traj.py:89-171) for B vehicle
 * poses against ONE global trajectory of M waypoints (columns as built by RefTrajectory.set_traj, ref_traj.py:67-86, plus the
 * unwrapped yaw).  H = traj_horizon (3..64), dt = traj_dt.  All arrays are device pointers.
 *   out_ref  [B][6][H]  x_ref, y_ref, psi_ref, v_ref (after the "three points from the current pose" splice, :158-170),
 *                       cdist_ref, curv_ref
 *   out_err  [B][3]     s0, e_y0, e_psi0          out_stop [B]  1 if the horizon reaches the end of the path (:153-155) */
int admpc_waypoints_batch(int device, int M, int H, double dt, int B,
                          const double* vel, const double* x, const double* y, const double* psi, const double* psi_unwrapped,
                          const double* cdist, const double* curv,
                          const double* X_init, const double* Y_init, const double* psi_init,
                          double* out_ref, double* out_err, int32_t* out_stop, void* stream);

const char* admpc_last_error(void);
const char* admpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADMPC_H */
