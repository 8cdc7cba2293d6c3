/*
 * admpc_oracle.h -- CPU oracle for the AD-MPC solve path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (ad_mpc_amd/, libadmpc.so) never does and has no CPU fallback.
 *
 * It shares nothing with the product but the plain-data AdmpcConfig struct layout.
 */
#ifndef ADMPC_ORACLE_H
#define ADMPC_ORACLE_H
#include "../include/admpc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* xdot = f(x,u,p)                       (ad_3d_optimizer.py:280-310) */
void oracle_f(const AdmpcConfig* c, const double* x, const double* u, double p, double* xdot);
/* Jx[7][7], Ju[7][2] row-major          (hand derivative of the same lines) */
void oracle_jac(const AdmpcConfig* c, const double* x, const double* u, double p, double* Jx, double* Ju);
/* one classic RK4 step of length h with forward sensitivities: phi[7], A[7][7], B[7][2]
 * (acados ERK, 4 stages, 1 step: acados_solver_sim_car.c:655-665) */
void oracle_rk4_sens(const AdmpcConfig* c, const double* x, const double* u, double p, double h,
                     double* phi, double* A, double* B);

/* Batched solve with the same argument meaning as admpc_solve_batch (host pointers).
 * nthreads>1 uses OpenMP when compiled with -fopenmp, otherwise it is ignored. */
int oracle_solve_batch(const AdmpcConfig* c, int B,
                       const double* x0, const double* yref, const double* yref_e, const double* p,
                       double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters,
                       int nthreads);

/* Receding-horizon shift of the iterate, as admpc_shift_batch (an option of this build; host pointers). */
int oracle_shift_batch(const AdmpcConfig* c, int B, double* xbar, double* ubar, const double* p, int rollout);

/* The QP of one RTI step solved for a single instance, returning everything the KKT checker in
 * tests/ needs: du[N][2], dx[N+1][7], the stage linearisation A[N][7][7], Bm[N][7][2], b[N][7]
 * and the inequality multipliers lam_u[N][2][4] (lower, upper, sl>=0, su>=0), lam_d[N][2]
 * (delta lower/upper, rows 0 and N unused), slacks sl[N][2], su[N][2]. */
int oracle_qp_debug(const AdmpcConfig* c,
                    const double* x0, const double* yref, const double* yref_e, double p,
                    const double* xbar, const double* ubar,
                    double* du, double* dx, double* A, double* Bm, double* b,
                    double* lam_u, double* lam_d, double* sl, double* su, int32_t* iters);

/* acados' four SQP stopping residuals (res_stat, res_eq, res_ineq, res_comp; restated from ocp_nlp_common.c:ocp_nlp_res_compute of the
 * acados commit the reference pins) for one instance at the iterate (xbar, ubar) with multipliers in the record layout of
 * include/admpc.h -- the checker of admpc_nlp_residuals_batch. */
int oracle_nlp_residuals(const AdmpcConfig* c, const double* x0, const double* yref, const double* yref_e, double p,
                         const double* xbar, const double* ubar, const double* pi, const double* ineq, double* res);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
