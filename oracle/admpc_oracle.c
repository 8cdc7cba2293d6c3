/*
 * admpc_oracle.c -- CPU restatement (plain scalar C, fp64) of the reference's AD-MPC solve path.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / timed CPU baseline.  The product never links or calls it.
 *
 * What it restates (paths relative to /root/reference/data_driven_mpc/ros_gp_mpc):
 *   model f(x,u,p)            src/ad_mpc/ad_3d_optimizer.py:280-310, constants src/ad_mpc/ad_3d.py:47-64
 *                             (twin: c_generated_code/sim_car_model/sim_car_expl_ode_fun.c:52-290)
 *   forward sensitivities     c_generated_code/sim_car_model/sim_car_expl_vde_forw.c:120 (CasADi AD of f)
 *   integrator                ERK4, one step per interval: c_generated_code/acados_solver_sim_car.c:655-665
 *   cost                      LINEAR_LS, Gauss-Newton, scaling Ts: ad_3d_optimizer.py:146-161,
 *                             acados_solver_sim_car.c:362-366,379-499
 *   constraints               soft u box (L1, zl=zu=10, Zl=Zu=0), hard delta box on stages 1..N-1,
 *                             x0 equality: ad_3d_optimizer.py:163-199, acados_solver_sim_car.c:505-605
 *   NLP step                  SQP_RTI, full step: ad_3d_optimizer.py:201-205, acados_solver_sim_car.c:647-681
 *   GP residual (config 3)    src/model_fitting/gp.py:81-138,446-471; wiring src/quad_mpc/quad_3d_optimizer.py:289-327
 *
 * Third-party arithmetic that is NOT in the reference tree: acados @91a01d4c (requirements.txt:1) with its
 * HPIPM/BLASFEO submodules.  acados condenses the stage QP and HPIPM solves it with a Mehrotra
 * primal-dual interior point method.  The QP is strictly convex in the inputs (R > 0), so its primal
 * minimiser is unique; this file solves the SAME QP with a stage-wise Riccati
 * factorisation inside a Mehrotra predictor-corrector IPM and is pinned by
 *   (1) tests/golden/shooting.json   : f, RK4, A, B from the reference's compiled CasADi C (oracle/_ref)
 *   (2) tests/golden/kat_sim_car_iterate.json : the converged acados iterate shipped in the reference
 *       (src/ad_mpc/sim_car_iterate.json), see oracle/make_golden.py.
 */
#include "admpc_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_LONG_DOUBLE
typedef long double real;
#define R_SIN sinl
#define R_COS cosl
#define R_EXP expl
#define R_FABS fabsl
#else
typedef double real;
#define R_SIN sin
#define R_COS cos
#define R_EXP exp
#define R_FABS fabs
#endif

#define NX ADMPC_NX
#define NU ADMPC_NU
#define NY ADMPC_NY
#define MAXN ADMPC_MAX_N
#define IPM_FLOOR ((real)1e-40)   /* keeps t, lam away from underflow; never active before convergence */

/* ------------------------------------------------------------------------------------------ */
/* model                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* mean and gradient of one SE GP with an anisotropic length scale over n_feat features (gp.py:81-138 kernel: sigma_f exp(-0.5
 * sum_d (z_d - x_d)^2 / l_d^2), :446-471 mean, :140-165 derivative); z_d = [x;u][feat[d]] */
static void gp_eval(const AdmpcGp* g, const real* x, const real* u, real* mu, real* dmu)
{
    real m = 0, d[ADMPC_GP_MAX_FEAT] = {0};
    real z[ADMPC_GP_MAX_FEAT];
    for (int k = 0; k < g->n_feat; ++k) z[k] = g->feat[k] < NX ? x[g->feat[k]] : u[g->feat[k] - NX];
    for (int i = 0; i < g->n_points; ++i) {
        real e = 0;
        for (int k = 0; k < g->n_feat; ++k) { real dz = z[k] - (real)g->Z[k][i]; e += dz * dz * (real)g->inv_l2[k]; }
        real ka = (real)g->sigma_f * R_EXP(-(real)0.5 * e) * (real)g->alpha[i];
        m += ka;
        for (int k = 0; k < g->n_feat; ++k) d[k] -= ka * (z[k] - (real)g->Z[k][i]) * (real)g->inv_l2[k];
    }
    *mu = m + (real)g->ymean;
    for (int k = 0; k < g->n_feat; ++k) dmu[k] = d[k];
}

static void model_f(const AdmpcConfig* c, const real* x, const real* u, real p, real* f)
{
    const real psi = x[2], vx = x[3], vy = x[4], r = x[5], dl = x[6];
    const real m = c->mass, LF = c->L_F, LR = c->L_R, Iz = c->Iz, Cf = c->Cf, Cr = c->Cr;
    const real cp = R_COS(psi), sp = R_SIN(psi);
    f[0] = vx * cp - vy * sp;                                   /* ad_3d_optimizer.py:281 */
    f[1] = vx * sp + vy * cp;                                   /* :284 */
    f[2] = r;                                                   /* :287 */
    const real v = vx + (real)1e-99;
    const real Ffy = 2 * Cf * (dl - (vy + LF * r) / v);         /* :290 */
    const real Fry = 2 * Cr * (LR * r - vy) / v;                /* :296 */
    const real dyn3 = u[0] - (1 / m) * Ffy * R_SIN(dl) + vy * r;            /* :291 */
    const real kin3 = u[0];                                                  /* :292 */
    const real dyn4 = (1 / m) * (Fry + Ffy * R_COS(dl)) - vx * r;           /* :298 */
    const real kin4 = (u[1] * vx + dl * u[0]) * LR / (LR + LF);             /* :299 */
    const real dyn5 = (1 / Iz) * (LF * Ffy * R_COS(dl) - LR * Fry);         /* :305 */
    const real kin5 = (u[1] * vx + dl * u[0]) / (LR + LF);                  /* :306 */
    f[3] = p * dyn3 + (1 - p) * kin3;                           /* :293 */
    f[4] = p * dyn4 + (1 - p) * kin4;                           /* :300 */
    f[5] = p * dyn5 + (1 - p) * kin5;                           /* :307 */
    f[6] = u[1];                                                /* :310 */
    for (int g = 0; g < c->n_gp; ++g) {                         /* quad_3d_optimizer.py:315 */
        const AdmpcGp* gp = &c->gp[g];
        real mu, dmu[ADMPC_GP_MAX_FEAT];
        gp_eval(gp, x, u, &mu, dmu);
        f[gp->out] += mu;
    }
}

/* analytic Jacobians of model_f (derived by hand from the lines cited above) */
static void model_jac(const AdmpcConfig* c, const real* x, const real* u, real p, real Jx[NX][NX], real Ju[NX][NU])
{
    const real psi = x[2], vx = x[3], vy = x[4], r = x[5], dl = x[6];
    const real m = c->mass, LF = c->L_F, LR = c->L_R, Iz = c->Iz, Cf = c->Cf, Cr = c->Cr;
    const real L = LR + LF;
    const real cp = R_COS(psi), sp = R_SIN(psi), cd = R_COS(dl), sd = R_SIN(dl);
    memset(Jx, 0, sizeof(real) * NX * NX);
    memset(Ju, 0, sizeof(real) * NX * NU);
    Jx[0][2] = -vx * sp - vy * cp; Jx[0][3] = cp; Jx[0][4] = -sp;
    Jx[1][2] = vx * cp - vy * sp;  Jx[1][3] = sp; Jx[1][4] = cp;
    Jx[2][5] = 1;
    const real v = vx + (real)1e-99;
    const real Ffy = 2 * Cf * (dl - (vy + LF * r) / v);
    const real Fry = 2 * Cr * (LR * r - vy) / v;
    /* gradients w.r.t. (vx, vy, r, delta) */
    const real gF[4] = { 2 * Cf * (vy + LF * r) / (v * v), -2 * Cf / v, -2 * Cf * LF / v, 2 * Cf };
    const real gR[4] = { -Fry / v, -2 * Cr / v, 2 * Cr * LR / v, 0 };
    real d3[4], d4[4], d5[4];
    for (int i = 0; i < 4; ++i) {
        d3[i] = -gF[i] * sd / m;
        d4[i] = (gR[i] + gF[i] * cd) / m;
        d5[i] = (LF * gF[i] * cd - LR * gR[i]) / Iz;
    }
    d3[1] += r;  d3[2] += vy; d3[3] += -Ffy * cd / m;
    d4[0] += -r; d4[2] += -vx; d4[3] += -Ffy * sd / m;
    d5[3] += -LF * Ffy * sd / Iz;
    const real k4[4] = { u[1] * LR / L, 0, 0, u[0] * LR / L };
    const real k5[4] = { u[1] / L, 0, 0, u[0] / L };
    for (int i = 0; i < 4; ++i) {
        Jx[3][3 + i] = p * d3[i];
        Jx[4][3 + i] = p * d4[i] + (1 - p) * k4[i];
        Jx[5][3 + i] = p * d5[i] + (1 - p) * k5[i];
    }
    Ju[3][0] = 1;
    Ju[4][0] = (1 - p) * dl * LR / L; Ju[4][1] = (1 - p) * vx * LR / L;
    Ju[5][0] = (1 - p) * dl / L;      Ju[5][1] = (1 - p) * vx / L;
    Ju[6][1] = 1;
    for (int g = 0; g < c->n_gp; ++g) {
        const AdmpcGp* gp = &c->gp[g];
        real mu, dmu[ADMPC_GP_MAX_FEAT];
        gp_eval(gp, x, u, &mu, dmu);
        for (int k = 0; k < gp->n_feat; ++k) {
            if (gp->feat[k] < NX) Jx[gp->out][gp->feat[k]] += dmu[k]; else Ju[gp->out][gp->feat[k] - NX] += dmu[k];
        }
    }
}

/* classic RK4 on the augmented system (x, Sx, Su), Sx(0)=I, Su(0)=0 */
static void rk4_sens(const AdmpcConfig* c, const real* x, const real* u, real p, real h,
                     real* phi, real A[NX][NX], real Bm[NX][NU])
{
    static const real cstage[4] = { 0, 0.5, 0.5, 1.0 };
    static const real wstage[4] = { 1.0 / 6, 2.0 / 6, 2.0 / 6, 1.0 / 6 };
    real kx[NX] = {0}, kS[NX][NX], kU[NX][NU];
    real accx[NX] = {0}, accS[NX][NX], accU[NX][NU];
    memset(kS, 0, sizeof kS); memset(kU, 0, sizeof kU);
    memset(accS, 0, sizeof accS); memset(accU, 0, sizeof accU);
    for (int s = 0; s < 4; ++s) {
        real X[NX], S[NX][NX], U[NX][NU];
        for (int i = 0; i < NX; ++i) {
            X[i] = x[i] + cstage[s] * h * kx[i];
            for (int j = 0; j < NX; ++j) S[i][j] = (i == j ? 1 : 0) + cstage[s] * h * kS[i][j];
            for (int j = 0; j < NU; ++j) U[i][j] = cstage[s] * h * kU[i][j];
        }
        real Jx[NX][NX], Ju[NX][NU];
        model_f(c, X, u, p, kx);
        model_jac(c, X, u, p, Jx, Ju);
        for (int i = 0; i < NX; ++i) {
            for (int j = 0; j < NX; ++j) {
                real a = 0;
                for (int l = 0; l < NX; ++l) a += Jx[i][l] * S[l][j];
                kS[i][j] = a;
            }
            for (int j = 0; j < NU; ++j) {
                real a = Ju[i][j];
                for (int l = 0; l < NX; ++l) a += Jx[i][l] * U[l][j];
                kU[i][j] = a;
            }
        }
        for (int i = 0; i < NX; ++i) {
            accx[i] += wstage[s] * kx[i];
            for (int j = 0; j < NX; ++j) accS[i][j] += wstage[s] * kS[i][j];
            for (int j = 0; j < NU; ++j) accU[i][j] += wstage[s] * kU[i][j];
        }
    }
    for (int i = 0; i < NX; ++i) {
        phi[i] = x[i] + h * accx[i];
        for (int j = 0; j < NX; ++j) A[i][j] = (i == j ? 1 : 0) + h * accS[i][j];
        for (int j = 0; j < NU; ++j) Bm[i][j] = h * accU[i][j];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* the QP of one RTI step and its interior point solve                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int N;
    real A[MAXN][NX][NX], B[MAXN][NX][NU], b[MAXN][NX];
    real Qd[NX], Rd[NU], Qe[NX];           /* Ts*q, Ts*r, W_e (all diagonal)                 */
    real q[MAXN + 1][NX], r[MAXN][NU];     /* gradients of the GN model at the iterate       */
    real dlu[MAXN][NU], duu[MAXN][NU];     /* lbu-ubar, ubu-ubar                              */
    real dld[MAXN], dud[MAXN];             /* delta bounds minus xbar[k][6], k=1..N-1         */
    real dx0[NX];
    real rho_l, rho_u;                     /* Ts*zl, Ts*zu                                    */
} StageQP;

typedef struct {
    real t[MAXN][NU][4], lam[MAXN][NU][4];   /* 0: lower soft, 1: upper soft, 2: sl>=0, 3: su>=0 */
    real td[MAXN][2], lamd[MAXN][2];         /* delta lower / upper, k = 1..N-1                   */
    real du[MAXN][NU], sl[MAXN][NU], su[MAXN][NU];
    real dx[MAXN + 1][NX], pi[MAXN][NX];     /* pi[k]: multiplier of dx[k+1] = A dx[k] + B du[k] + b */
} IpmState;

typedef struct {
    real K[MAXN][NU][NX];
    real Li[MAXN][3];       /* inverse of Huu: [i00 i01 i11] */
    real P[MAXN + 1][NX][NX];
} RicFactor;

/* Backward Riccati recursion for the matrices.  Rt/Qt66 = barrier-augmented diagonals. */
static void riccati_factor(const StageQP* qp, const real Rt[][NU], const real* Qt66, RicFactor* F)
{
    const int N = qp->N;
    memset(F->P[N], 0, sizeof(F->P[N]));
    for (int i = 0; i < NX; ++i) F->P[N][i][i] = qp->Qe[i];
    for (int k = N - 1; k >= 0; --k) {
        real (*P)[NX] = F->P[k + 1];
        real PA[NX][NX], PB[NX][NU];
        for (int i = 0; i < NX; ++i) {
            for (int j = 0; j < NX; ++j) { real a = 0; for (int l = 0; l < NX; ++l) a += P[i][l] * qp->A[k][l][j]; PA[i][j] = a; }
            for (int j = 0; j < NU; ++j) { real a = 0; for (int l = 0; l < NX; ++l) a += P[i][l] * qp->B[k][l][j]; PB[i][j] = a; }
        }
        real Huu[NU][NU], Hux[NU][NX], Hxx[NX][NX];
        for (int i = 0; i < NU; ++i) {
            for (int j = 0; j < NU; ++j) { real a = 0; for (int l = 0; l < NX; ++l) a += qp->B[k][l][i] * PB[l][j]; Huu[i][j] = a; }
            for (int j = 0; j < NX; ++j) { real a = 0; for (int l = 0; l < NX; ++l) a += qp->B[k][l][i] * PA[l][j]; Hux[i][j] = a; }
            Huu[i][i] += Rt[k][i];
        }
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j) { real a = 0; for (int l = 0; l < NX; ++l) a += qp->A[k][l][i] * PA[l][j]; Hxx[i][j] = a; }
        for (int i = 0; i < NX; ++i) Hxx[i][i] += (i == 6 && k >= 1) ? Qt66[k] : qp->Qd[i];
        /* 2x2 inverse */
        real det = Huu[0][0] * Huu[1][1] - Huu[0][1] * Huu[1][0];
        real i00 = Huu[1][1] / det, i01 = -Huu[0][1] / det, i11 = Huu[0][0] / det;
        F->Li[k][0] = i00; F->Li[k][1] = i01; F->Li[k][2] = i11;
        for (int j = 0; j < NX; ++j) {
            F->K[k][0][j] = -(i00 * Hux[0][j] + i01 * Hux[1][j]);
            F->K[k][1][j] = -(i01 * Hux[0][j] + i11 * Hux[1][j]);
        }
        /* P_k = Hxx + Hxu K, symmetrised */
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j <= i; ++j) {
                real a = Hxx[i][j] + Hux[0][i] * F->K[k][0][j] + Hux[1][i] * F->K[k][1][j];
                real bsym = Hxx[j][i] + Hux[0][j] * F->K[k][0][i] + Hux[1][j] * F->K[k][1][i];
                real s = (real)0.5 * (a + bsym);
                F->P[k][i][j] = s; F->P[k][j][i] = s;
            }
    }
}

/* Vector part for the Newton step: backward sweep of the gradient, forward roll-out from ddx[0]=0.
 *   minimise sum 1/2 dz'H~dz + gx'ddx + gu'ddu   s.t.  ddx[k+1] = A ddx[k] + B ddu[k] + beq[k]
 * also returns the multiplier step dpi[k] = P[k+1] ddx[k+1] + p[k+1]. */
static void riccati_solve(const StageQP* qp, const RicFactor* F, const real gx[][NX], const real gu[][NU], const real beq[][NX],
                          real ddu[][NU], real ddx[][NX], real dpi[][NX])
{
    const int N = qp->N;
    real pv[MAXN + 1][NX], kff[MAXN][NU];
    for (int i = 0; i < NX; ++i) pv[N][i] = gx[N][i];
    for (int k = N - 1; k >= 0; --k) {
        const real (*P)[NX] = F->P[k + 1];
        real w[NX], hx[NX], hu[NU];
        for (int i = 0; i < NX; ++i) { real a = pv[k + 1][i]; for (int l = 0; l < NX; ++l) a += P[i][l] * beq[k][l]; w[i] = a; }
        for (int j = 0; j < NU; ++j) { real a = gu[k][j]; for (int l = 0; l < NX; ++l) a += qp->B[k][l][j] * w[l]; hu[j] = a; }
        for (int j = 0; j < NX; ++j) { real a = gx[k][j]; for (int l = 0; l < NX; ++l) a += qp->A[k][l][j] * w[l]; hx[j] = a; }
        kff[k][0] = -(F->Li[k][0] * hu[0] + F->Li[k][1] * hu[1]);
        kff[k][1] = -(F->Li[k][1] * hu[0] + F->Li[k][2] * hu[1]);
        for (int j = 0; j < NX; ++j) pv[k][j] = hx[j] + F->K[k][0][j] * hu[0] + F->K[k][1][j] * hu[1];
    }
    for (int i = 0; i < NX; ++i) ddx[0][i] = 0;
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j) { real a = kff[k][j]; for (int l = 0; l < NX; ++l) a += F->K[k][j][l] * ddx[k][l]; ddu[k][j] = a; }
        for (int i = 0; i < NX; ++i) {
            real a = beq[k][i];
            for (int l = 0; l < NX; ++l) a += qp->A[k][i][l] * ddx[k][l];
            for (int l = 0; l < NU; ++l) a += qp->B[k][i][l] * ddu[k][l];
            ddx[k + 1][i] = a;
        }
        for (int i = 0; i < NX; ++i) { real a = pv[k + 1][i]; for (int l = 0; l < NX; ++l) a += F->P[k + 1][i][l] * ddx[k + 1][l]; dpi[k][i] = a; }
    }
}

typedef struct {
    /* residuals */
    real ru[MAXN][NU], rx[MAXN + 1][NX], rsl[MAXN][NU], rsu[MAXN][NU], req[MAXN][NX];
    real rd[MAXN][NU][4], rdd[MAXN][2];
    real rc[MAXN][NU][4], rcd[MAXN][2];          /* complementarity rhs (lam*t, + corrector terms) */
    /* reduced system */
    real Rt[MAXN][NU], Qt66[MAXN], gu[MAXN][NU], gx[MAXN + 1][NX], e1[MAXN][NU], e2[MAXN][NU];
    /* steps */
    real ddu[MAXN][NU], ddx[MAXN + 1][NX], dpi[MAXN][NX], dsl[MAXN][NU], dsu[MAXN][NU];
    real dt[MAXN][NU][4], dlam[MAXN][NU][4], dtd[MAXN][2], dlamd[MAXN][2];
} IpmWork;

/* all linear residuals at the current iterate; returns their max-norm.  *m_ineq (may be NULL): max-norm over the inequality rows the
 * stopping test re-evaluates (bound pairs, slack-penalty rows, steering rows); *m_stat: over the stationarity rows (ru, rx). */
static real ipm_residuals(const StageQP* qp, const IpmState* s, IpmWork* w, real* m_ineq, real* m_stat)
{
    const int N = qp->N;
    real m = 0, mi = 0, ms = 0;
#define UPD1(acc, v) do { real a_ = R_FABS(v); if (a_ > acc || !(a_ == a_)) acc = a_; } while (0)
#define UPD(v) UPD1(m, v)
#define UPDI(v) do { UPD1(m, v); UPD1(mi, v); } while (0)
#define UPDS(v) do { UPD1(m, v); UPD1(ms, v); } while (0)
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j) {
            real a = qp->Rd[j] * s->du[k][j] + qp->r[k][j] - s->lam[k][j][0] + s->lam[k][j][1];
            for (int l = 0; l < NX; ++l) a += qp->B[k][l][j] * s->pi[k][l];
            w->ru[k][j] = a; UPDS(a);
            w->rsl[k][j] = qp->rho_l - s->lam[k][j][0] - s->lam[k][j][2]; UPDI(w->rsl[k][j]);
            w->rsu[k][j] = qp->rho_u - s->lam[k][j][1] - s->lam[k][j][3]; UPDI(w->rsu[k][j]);
            w->rd[k][j][0] = s->du[k][j] + s->sl[k][j] - qp->dlu[k][j] - s->t[k][j][0];
            w->rd[k][j][1] = -s->du[k][j] + s->su[k][j] + qp->duu[k][j] - s->t[k][j][1];
            w->rd[k][j][2] = s->sl[k][j] - s->t[k][j][2];
            w->rd[k][j][3] = s->su[k][j] - s->t[k][j][3];
            UPDI(w->rd[k][j][0]); UPDI(w->rd[k][j][1]); UPD(w->rd[k][j][2]); UPD(w->rd[k][j][3]);
        }
        for (int i = 0; i < NX; ++i) {
            real a = qp->b[k][i] - s->dx[k + 1][i];
            for (int l = 0; l < NX; ++l) a += qp->A[k][i][l] * s->dx[k][l];
            for (int l = 0; l < NU; ++l) a += qp->B[k][i][l] * s->du[k][l];
            w->req[k][i] = a; UPD(a);
        }
        if (k >= 1) {
            for (int i = 0; i < NX; ++i) {
                real a = qp->Qd[i] * s->dx[k][i] + qp->q[k][i] - s->pi[k - 1][i];
                for (int l = 0; l < NX; ++l) a += qp->A[k][l][i] * s->pi[k][l];
                w->rx[k][i] = a;
            }
            w->rx[k][6] += -s->lamd[k][0] + s->lamd[k][1];
            for (int i = 0; i < NX; ++i) UPDS(w->rx[k][i]);
            w->rdd[k][0] = s->dx[k][6] - qp->dld[k] - s->td[k][0]; UPDI(w->rdd[k][0]);
            w->rdd[k][1] = qp->dud[k] - s->dx[k][6] - s->td[k][1]; UPDI(w->rdd[k][1]);
        } else {
            for (int i = 0; i < NX; ++i) w->rx[0][i] = 0;
        }
    }
    for (int i = 0; i < NX; ++i) { w->rx[N][i] = qp->Qe[i] * s->dx[N][i] + qp->q[N][i] - s->pi[N - 1][i]; UPDS(w->rx[N][i]); }
#undef UPD
#undef UPDI
#undef UPDS
#undef UPD1
    if (m_ineq) *m_ineq = mi;
    if (m_stat) *m_stat = ms;
    return m;
}

/* eliminate slacks / inequality multipliers: barrier-augmented diagonals and gradients */
static void ipm_reduce(const StageQP* qp, const IpmState* s, IpmWork* w, int hessian_too)
{
    const int N = qp->N;
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) w->gx[k][i] = w->rx[k][i];
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j) {
            const real* t = s->t[k][j]; const real* l = s->lam[k][j]; const real* rc = w->rc[k][j]; const real* rd = w->rd[k][j];
            real G0 = l[0] / t[0], G1 = l[1] / t[1], G2 = l[2] / t[2], G3 = l[3] / t[3];
            real e1 = w->rsl[k][j] + rc[0] / t[0] + rc[2] / t[2] + G0 * rd[0] + G2 * rd[2];
            real e2 = w->rsu[k][j] + rc[1] / t[1] + rc[3] / t[3] + G1 * rd[1] + G3 * rd[3];
            real etal = rc[0] / t[0] + G0 * rd[0] - G0 * e1 / (G0 + G2);
            real etau = -rc[1] / t[1] - G1 * rd[1] + G1 * e2 / (G1 + G3);
            if (hessian_too) w->Rt[k][j] = qp->Rd[j] + G0 * G2 / (G0 + G2) + G1 * G3 / (G1 + G3);
            w->gu[k][j] = w->ru[k][j] + etal + etau;
            w->e1[k][j] = e1; w->e2[k][j] = e2;
        }
        if (k >= 1) {
            real G5 = s->lamd[k][0] / s->td[k][0], G6 = s->lamd[k][1] / s->td[k][1];
            if (hessian_too) w->Qt66[k] = qp->Qd[6] + G5 + G6;
            w->gx[k][6] += (w->rcd[k][0] / s->td[k][0] + G5 * w->rdd[k][0]) - (w->rcd[k][1] / s->td[k][1] + G6 * w->rdd[k][1]);
        } else if (hessian_too) w->Qt66[k] = qp->Qd[6];
    }
}

/* recover the slack / t / lam steps from (ddu, ddx) */
static void ipm_expand(const StageQP* qp, const IpmState* s, IpmWork* w)
{
    const int N = qp->N;
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j) {
            const real* t = s->t[k][j]; const real* l = s->lam[k][j]; const real* rc = w->rc[k][j]; const real* rd = w->rd[k][j];
            real G[4] = { l[0] / t[0], l[1] / t[1], l[2] / t[2], l[3] / t[3] };
            real u = w->ddu[k][j];
            real dsl = -(w->e1[k][j] + G[0] * u) / (G[0] + G[2]);
            real dsu = -(w->e2[k][j] - G[1] * u) / (G[1] + G[3]);
            real dtv[4] = { u + dsl + rd[0], -u + dsu + rd[1], dsl + rd[2], dsu + rd[3] };
            for (int i = 0; i < 4; ++i) { w->dt[k][j][i] = dtv[i]; w->dlam[k][j][i] = -rc[i] / t[i] - G[i] * dtv[i]; }
            w->dsl[k][j] = dsl; w->dsu[k][j] = dsu;
        }
        if (k >= 1) {
            real G5 = s->lamd[k][0] / s->td[k][0], G6 = s->lamd[k][1] / s->td[k][1];
            real x6 = w->ddx[k][6];
            w->dtd[k][0] = x6 + w->rdd[k][0];  w->dlamd[k][0] = -w->rcd[k][0] / s->td[k][0] - G5 * w->dtd[k][0];
            w->dtd[k][1] = -x6 + w->rdd[k][1]; w->dlamd[k][1] = -w->rcd[k][1] / s->td[k][1] - G6 * w->dtd[k][1];
        }
    }
}

static real ipm_max_step(const StageQP* qp, const IpmState* s, const IpmWork* w)
{
    const int N = qp->N;
    real a = 1;
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j)
            for (int i = 0; i < 4; ++i) {
                if (w->dt[k][j][i] < 0)   { real c = -s->t[k][j][i] / w->dt[k][j][i];     if (c < a) a = c; }
                if (w->dlam[k][j][i] < 0) { real c = -s->lam[k][j][i] / w->dlam[k][j][i]; if (c < a) a = c; }
            }
        if (k >= 1)
            for (int i = 0; i < 2; ++i) {
                if (w->dtd[k][i] < 0)   { real c = -s->td[k][i] / w->dtd[k][i];     if (c < a) a = c; }
                if (w->dlamd[k][i] < 0) { real c = -s->lamd[k][i] / w->dlamd[k][i]; if (c < a) a = c; }
            }
    }
    return a;
}

/* cold start (qp_solver_warm_start 0, sim_car_acados_ocp.json:885): zero step, slacks at thr */
static void ipm_cold_start(const AdmpcConfig* c, const StageQP* qp, IpmState* s)
{
    const int N = qp->N;
    const real thr = c->ipm_thr0, mu0 = c->ipm_mu0;
    memset(s, 0, sizeof *s);
    /* primal start: zero input step, states rolled out through the linearised dynamics, so every
     * iterate satisfies dx[k+1] = A dx[k] + B du[k] + b up to rounding */
    for (int i = 0; i < NX; ++i) s->dx[0][i] = qp->dx0[i];
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < NX; ++i) {
            real a = qp->b[k][i];
            for (int l = 0; l < NX; ++l) a += qp->A[k][i][l] * s->dx[k][l];
            s->dx[k + 1][i] = a;
        }
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j) {
            s->sl[k][j] = thr; s->su[k][j] = thr;
            real r0[4] = { thr - qp->dlu[k][j], thr + qp->duu[k][j], thr, thr };
            for (int i = 0; i < 4; ++i) { s->t[k][j][i] = r0[i] > thr ? r0[i] : thr; s->lam[k][j][i] = mu0 / s->t[k][j][i]; }
        }
        s->td[k][0] = s->td[k][1] = 1;
        if (k >= 1) {
            real r0[2] = { s->dx[k][6] - qp->dld[k], qp->dud[k] - s->dx[k][6] };
            for (int i = 0; i < 2; ++i) { s->td[k][i] = r0[i] > thr ? r0[i] : thr; s->lamd[k][i] = mu0 / s->td[k][i]; }
        }
    }
}

/* scripts/ only (predictor studies): per-iteration trace of one solve -- mu at the start of the iteration, the step length taken */
static _Thread_local double* g_trace = 0; static _Thread_local int g_trace_cap = 0;

/* Mehrotra predictor-corrector primal-dual IPM in residual (Newton-step) form.
 * returns number of IPM iterations, negative on numerical failure */
static int ipm_solve(const AdmpcConfig* c, const StageQP* qp, IpmState* s, IpmWork* w, RicFactor* F)
{
    const int N = qp->N;
    const int n_ineq = 8 * N + 2 * (N - 1);
    const real mu0 = c->ipm_mu0;
    int warmed = 0;                      /* the interior point starts from the trial's minimiser (cfg.ipm_warm_thr) */
    int cons = 0;                        /* fallback mode (cfg.ipm_fallback_iter): no second-order corrector term */
    ipm_cold_start(c, qp, s);
    if (c->ipm_try_unconstrained != 0) {
        /* Newton step of the QP WITHOUT its inequalities from the start point (du = 0, rolled-out dx, pi = 0): the exact
         * minimiser of the equality-constrained QP.  If it respects the input box and the delta box it is the solution of
         * the full QP (no bound active, slacks zero, L1 penalty idle) and the interior point is not needed (iters = 0). */
        for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i)
            w->gx[k][i] = (k == 0) ? 0 : ((k < N ? qp->Qd[i] : qp->Qe[i]) * s->dx[k][i] + qp->q[k][i]);
        for (int k = 0; k < N; ++k) {
            for (int j = 0; j < NU; ++j) { w->gu[k][j] = qp->r[k][j]; w->Rt[k][j] = qp->Rd[j]; }
            w->Qt66[k] = qp->Qd[6];
            for (int i = 0; i < NX; ++i) w->req[k][i] = 0;
        }
        riccati_factor(qp, (const real (*)[NU])w->Rt, w->Qt66, F);
        riccati_solve(qp, F, (const real (*)[NX])w->gx, (const real (*)[NU])w->gu, (const real (*)[NX])w->req, w->ddu, w->ddx, w->dpi);
        int ok = 1;
        for (int k = 0; k < N && ok; ++k) {
            for (int j = 0; j < NU; ++j) { real v = w->ddu[k][j]; if (!(v >= qp->dlu[k][j] && v <= qp->duu[k][j])) ok = 0; }
            if (k >= 1) { real v = s->dx[k][6] + w->ddx[k][6]; if (!(v >= qp->dld[k] && v <= qp->dud[k])) ok = 0; }
        }
        if (ok) {
            for (int k = 0; k < N; ++k) {
                for (int j = 0; j < NU; ++j) {
                    s->du[k][j] = w->ddu[k][j]; s->sl[k][j] = s->su[k][j] = 0;
                    s->lam[k][j][0] = s->lam[k][j][1] = 0;                     /* box multipliers: inactive              */
                    s->lam[k][j][2] = qp->rho_l; s->lam[k][j][3] = qp->rho_u;  /* sl, su >= 0 hold the L1 penalty weight */
                    s->t[k][j][0] = s->du[k][j] - qp->dlu[k][j]; s->t[k][j][1] = qp->duu[k][j] - s->du[k][j];
                    s->t[k][j][2] = s->t[k][j][3] = 0;
                }
                for (int i = 0; i < NX; ++i) { s->dx[k + 1][i] += w->ddx[k + 1][i]; s->pi[k][i] = w->dpi[k][i]; }
                s->lamd[k][0] = s->lamd[k][1] = 0;
            }
            for (int k = 1; k < N; ++k) { s->td[k][0] = s->dx[k][6] - qp->dld[k]; s->td[k][1] = qp->dud[k] - s->dx[k][6]; }   /* slacks of the idle delta box */
            return 0;
        }
        if (c->ipm_warm_thr > 0) {
            warmed = 1;
            /* warm start (cfg.ipm_warm_thr): the interior point starts from that minimiser instead of from the zero step.
             * Inputs, states and dynamics multipliers are the minimiser's; a violated input bound is absorbed by its slack
             * (the bounds are soft), a violated delta bound is left as a primal residual (residual form of the method). */
            const real thw = c->ipm_warm_thr;
            for (int k = 0; k < N; ++k) {
                for (int j = 0; j < NU; ++j) {
                    const real v = w->ddu[k][j];
                    const real vl = qp->dlu[k][j] - v, vu = v - qp->duu[k][j];
                    s->du[k][j] = v;
                    s->sl[k][j] = (vl > 0 ? vl : 0) + thw; s->su[k][j] = (vu > 0 ? vu : 0) + thw;
                    real r0[4] = { v + s->sl[k][j] - qp->dlu[k][j], -v + s->su[k][j] + qp->duu[k][j], s->sl[k][j], s->su[k][j] };
                    for (int i = 0; i < 4; ++i) { s->t[k][j][i] = r0[i] > thw ? r0[i] : thw; s->lam[k][j][i] = mu0 / s->t[k][j][i]; }
                }
                for (int i = 0; i < NX; ++i) { s->dx[k + 1][i] += w->ddx[k + 1][i]; s->pi[k][i] = w->dpi[k][i]; }
            }
            for (int k = 1; k < N; ++k) {
                real r0[2] = { s->dx[k][6] - qp->dld[k], qp->dud[k] - s->dx[k][6] };
                for (int i = 0; i < 2; ++i) { s->td[k][i] = r0[i] > thw ? r0[i] : thw; s->lamd[k][i] = mu0 / s->td[k][i]; }
            }
        }
    }
    int it;
    real rmax_prev = 0, step = 1e300;   /* step: max-norm of the last applied input step */
    real alpha_prev = 1;                /* step length of the previous iteration (centring safeguard, admpc.h) */
    /* Stopping test on the linear residuals: every linear residual of a Newton iteration in residual form shrinks by exactly
     * (1 - alpha) per step.  The stationarity rows (ru, rx: they carry the dynamics multipliers) enter the test through that law --
     * rstat starts at their max-norm at the start point and is multiplied by (1 - alpha) per step -- the inequality rows are
     * re-evaluated.  Re-evaluating the stationarity rows as well puts their accumulated rounding (1e-10 .. 1e-9 at N >= 80) next to
     * tol_res = 1e-9, and whether an instance on its rounding floor stops one iteration earlier or later then depends on the order
     * of summation (the device kernels do not iterate the dynamics multipliers at all: rowqp_core.h).  The re-evaluated value is
     * still formed every iteration: a non-finite entry anywhere is a QP failure. */
    real rstat = -1;                    /* < 0: (re)start -- take the start point's value */
    for (it = 0; it < c->ipm_iter_max + (cons ? (int)c->ipm_fallback_iter : 0); ++it) {      /* the fallback gets a full budget of its own */
        real mu, cmax;
    residuals:
        mu = 0; cmax = 0;
        for (int k = 0; k < N; ++k) {
            for (int j = 0; j < NU; ++j) for (int i = 0; i < 4; ++i) { real v = s->t[k][j][i] * s->lam[k][j][i]; w->rc[k][j][i] = v; mu += v; if (v > cmax) cmax = v; }
            if (k >= 1) for (int i = 0; i < 2; ++i) { real v = s->td[k][i] * s->lamd[k][i]; w->rcd[k][i] = v; mu += v; if (v > cmax) cmax = v; }
        }
        mu /= n_ineq;
        real rineq, rstat_now;
        real rall = ipm_residuals(qp, s, w, &rineq, &rstat_now);
        if (!(mu == mu) || !(rall == rall)) return -1;
        if (rstat < 0) rstat = rstat_now;
        real rmax = rineq > rstat ? rineq : rstat;
        /* converged, or complementarity reached and the linear residuals sit on their rounding floor */
        if (cmax <= c->ipm_tol_comp && step <= c->ipm_tol_step &&
            (rmax <= c->ipm_tol_res || (it > 0 && rmax > (real)0.1 * rmax_prev && rmax <= (real)ADMPC_IPM_FLOOR_CAP * c->ipm_tol_res))) break;
        rmax_prev = rmax;
        if (!cons && c->ipm_fallback_iter > 0 && it >= (int)c->ipm_fallback_iter) {
            /* still iterating: most likely a limit cycle of the centring heuristic.  Start over, finish with plain predictor-centring steps */
            cons = 1; warmed = 0;
            ipm_cold_start(c, qp, s);
            alpha_prev = 1; step = 1e300; rmax_prev = 0; rstat = -1;
            goto residuals;
        }

        /* predictor (affine scaling direction) */
        ipm_reduce(qp, s, w, 1);
        riccati_factor(qp, (const real (*)[NU])w->Rt, w->Qt66, F);
        riccati_solve(qp, F, (const real (*)[NX])w->gx, (const real (*)[NU])w->gu, (const real (*)[NX])w->req, w->ddu, w->ddx, w->dpi);
        ipm_expand(qp, s, w);
        real a_aff = ipm_max_step(qp, s, w);
        real mu_aff = 0;
        for (int k = 0; k < N; ++k) {
            for (int j = 0; j < NU; ++j) for (int i = 0; i < 4; ++i)
                mu_aff += (s->t[k][j][i] + a_aff * w->dt[k][j][i]) * (s->lam[k][j][i] + a_aff * w->dlam[k][j][i]);
            if (k >= 1) for (int i = 0; i < 2; ++i)
                mu_aff += (s->td[k][i] + a_aff * w->dtd[k][i]) * (s->lamd[k][i] + a_aff * w->dlamd[k][i]);
        }
        mu_aff /= n_ineq;
        real sigma = mu_aff / mu; sigma = sigma * sigma * sigma;
        if (alpha_prev < (real)ADMPC_IPM_BLOCKED_STEP) sigma = 1;       /* blocked step: centre (breaks the method's limit cycles) */
        real smu = sigma * mu;                                          /* centring target, never below MU_FLOOR * tol_comp (admpc.h) */
        if (smu < (real)ADMPC_IPM_MU_FLOOR * (real)c->ipm_tol_comp) smu = (real)ADMPC_IPM_MU_FLOOR * (real)c->ipm_tol_comp;
        /* corrector: centring + second-order term (dropped in fallback mode) */
        const real w2 = cons ? 0 : 1;
        for (int k = 0; k < N; ++k) {
            for (int j = 0; j < NU; ++j) for (int i = 0; i < 4; ++i)
                w->rc[k][j][i] = s->t[k][j][i] * s->lam[k][j][i] + w2 * (w->dt[k][j][i] * w->dlam[k][j][i]) - smu;
            if (k >= 1) for (int i = 0; i < 2; ++i)
                w->rcd[k][i] = s->td[k][i] * s->lamd[k][i] + w2 * (w->dtd[k][i] * w->dlamd[k][i]) - smu;
        }
        ipm_reduce(qp, s, w, 0);
        riccati_solve(qp, F, (const real (*)[NX])w->gx, (const real (*)[NU])w->gu, (const real (*)[NX])w->req, w->ddu, w->ddx, w->dpi);
        ipm_expand(qp, s, w);
        real a_max = ipm_max_step(qp, s, w);
        real tau = 1 - mu_aff; if (tau < (real)0.995) tau = (real)0.995; if (tau > (real)0.999999) tau = (real)0.999999;
        real alpha = tau * a_max; if (alpha > 1) alpha = 1;
        if (it == 0 && warmed && alpha < (real)c->ipm_warm_restart) {
            /* the first step from the warm start is blocked (cfg.ipm_warm_restart): start over from the cold start; the iteration counts */
            warmed = 0;
            ipm_cold_start(c, qp, s);
            alpha_prev = 1; step = 1e300; rmax_prev = 0; rstat = -1;
            continue;
        }
        if (g_trace && it < g_trace_cap) { g_trace[2 * it] = (double)mu; g_trace[2 * it + 1] = (double)alpha; }
        alpha_prev = alpha;
        rstat = (1 - alpha) * rstat;
        step = 0;
        for (int k = 0; k < N; ++k) {
            for (int j = 0; j < NU; ++j) {
                real sj = R_FABS(alpha * w->ddu[k][j]); if (sj > step) step = sj;
                for (int i = 0; i < 4; ++i) {
                    s->t[k][j][i] += alpha * w->dt[k][j][i]; s->lam[k][j][i] += alpha * w->dlam[k][j][i];
                    if (s->t[k][j][i] < IPM_FLOOR) s->t[k][j][i] = IPM_FLOOR;
                    if (s->lam[k][j][i] < IPM_FLOOR) s->lam[k][j][i] = IPM_FLOOR;
                }
                s->du[k][j] += alpha * w->ddu[k][j];
                s->sl[k][j] += alpha * w->dsl[k][j];
                s->su[k][j] += alpha * w->dsu[k][j];
            }
            for (int i = 0; i < NX; ++i) { s->dx[k + 1][i] += alpha * w->ddx[k + 1][i]; s->pi[k][i] += alpha * w->dpi[k][i]; }
            if (k >= 1) for (int i = 0; i < 2; ++i) {
                s->td[k][i] += alpha * w->dtd[k][i]; s->lamd[k][i] += alpha * w->dlamd[k][i];
                if (s->td[k][i] < IPM_FLOOR) s->td[k][i] = IPM_FLOOR;
                if (s->lamd[k][i] < IPM_FLOOR) s->lamd[k][i] = IPM_FLOOR;
            }
        }
    }
    return it;
}

/* ------------------------------------------------------------------------------------------ */
/* one RTI step of one instance                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    StageQP qp; IpmState st; IpmWork wk; RicFactor F; real dx[MAXN + 1][NX];
    real pia[MAXN + 1][NX];      /* SQP mode with a tolerance: dynamics multipliers of the last QP by the adjoint recursion; [N]: of dx_0 = x0 - xbar_0 */
} Workspace;

/* H1-H3: linearisation of the OCP at the iterate -> the stage QP of this step */
static void build_qp(const AdmpcConfig* c, Workspace* W, const double* x0, const double* yref, const double* yref_e,
                     double p, const double* xbar, const double* ubar)
{
    const int N = c->N;
    StageQP* qp = &W->qp;
    qp->N = N;
    qp->rho_l = c->Ts * c->zl; qp->rho_u = c->Ts * c->zu;
    for (int i = 0; i < NX; ++i) { qp->Qd[i] = c->Ts * c->W[i]; qp->Qe[i] = c->We[i]; }
    for (int j = 0; j < NU; ++j) qp->Rd[j] = c->Ts * c->W[NX + j];
    for (int k = 0; k < N; ++k) {                                           /* H1: shooting */
        real xk[NX], uk[NU], phi[NX];
        for (int i = 0; i < NX; ++i) xk[i] = xbar[k * NX + i];
        for (int j = 0; j < NU; ++j) uk[j] = ubar[k * NU + j];
        rk4_sens(c, xk, uk, (real)p, (real)c->Ts, phi, qp->A[k], qp->B[k]);
        for (int i = 0; i < NX; ++i) qp->b[k][i] = phi[i] - (real)xbar[(k + 1) * NX + i];
        for (int i = 0; i < NX; ++i) qp->q[k][i] = qp->Qd[i] * (xk[i] - (real)yref[k * NY + i]);      /* H2 */
        for (int j = 0; j < NU; ++j) {
            qp->r[k][j] = qp->Rd[j] * (uk[j] - (real)yref[k * NY + NX + j]);
            qp->dlu[k][j] = (real)c->lbu[j] - uk[j]; qp->duu[k][j] = (real)c->ubu[j] - uk[j];          /* H3 */
        }
        qp->dld[k] = (real)c->lbx_delta - xk[6]; qp->dud[k] = (real)c->ubx_delta - xk[6];
    }
    for (int i = 0; i < NX; ++i) {
        qp->q[N][i] = qp->Qe[i] * ((real)xbar[N * NX + i] - (real)yref_e[i]);
        qp->dx0[i] = (real)x0[i] - (real)xbar[i];
    }
}

/* SQP mode with a tolerance (reference solver_type "SQP", create_ros_ad_mpc.py:47-51; the tolerances are acados' defaults,
 * acados_models/sim_car_acados_ocp.json:870-873 nlp_solver_tol_{comp,eq,ineq,stat} = 1e-6).  acados is not vendored in the
 * reference (requirements.txt:1 pins github.com/acados/acados at commit 91a01d4c8db1; acados/ocp_nlp/ocp_nlp_sqp.c and
 * ocp_nlp_common.c:ocp_nlp_res_compute there, restated from their published algorithm): its SQP loop is
 *     for (iter < nlp_solver_max_iter) { linearise at the iterate; residuals of the NLP's KKT system with the iterate's
 *         multipliers; all four inf-norms <= their tolerance -> ACADOS_SUCCESS; solve the QP; full step }  -> ACADOS_MAXITER
 * with  res_stat = || grad of the Lagrangian ||, res_eq = || shooting defects ||, res_ineq = || constraint + slack t ||,
 * res_comp = || lam .* t ||.  With FULL_CONDENSING_HPIPM the QP solver returns no dynamics multipliers of its own: the
 * expansion of the condensed solution recovers them by the adjoint recursion, which is what adjoint_multipliers() does (and what
 * kernel R's sweep_adjoint does on the device).
 * adjoint_multipliers: call after the full step, with the QP (old linearisation) still in W->qp and the new iterate in xbar. */
static void adjoint_multipliers(const AdmpcConfig* c, Workspace* W, const double* yref, const double* yref_e, const double* xbar)
{
    const int N = c->N;
    const StageQP* qp = &W->qp; const IpmState* s = &W->st;
    real lam[NX];
    for (int i = 0; i < NX; ++i) lam[i] = qp->Qe[i] * ((real)xbar[N * NX + i] - (real)yref_e[i]);
    for (int k = N - 1; k >= 0; --k) {
        for (int i = 0; i < NX; ++i) W->pia[k][i] = lam[i];                 /* pi_k multiplies dx_{k+1} = A dx_k + B du_k + b_k */
        real nx[NX];
        for (int i = 0; i < NX; ++i) {
            real a = qp->Qd[i] * ((real)xbar[k * NX + i] - (real)yref[k * NY + i]);
            if (i == 6 && k >= 1) a += -s->lamd[k][0] + s->lamd[k][1];
            for (int l = 0; l < NX; ++l) a += qp->A[k][l][i] * lam[l];
            nx[i] = a;
        }
        for (int i = 0; i < NX; ++i) lam[i] = nx[i];
    }
    for (int i = 0; i < NX; ++i) W->pia[N][i] = lam[i];                     /* multiplier of the initial-state equality */
}

/* the four residuals at the iterate the QP in W->qp was linearised at (zero step), with the multipliers and slacks the previous QP
 * left in W->st / W->pia; the slack variables are read from the slacks of their own bounds (sl = t[2], su = t[3]: the device record
 * holds nothing else).  res: stat, eq, ineq, comp */
static void nlp_residuals(const Workspace* W, real res[4])
{
    const StageQP* qp = &W->qp; const IpmState* s = &W->st;
    const int N = qp->N;
    real rs = 0, re = 0, ri = 0, rc = 0;
#define UPD1(acc, v) do { real a_ = R_FABS(v); if (a_ > acc || !(a_ == a_)) acc = a_; } while (0)
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j) {
            const real* t = s->t[k][j]; const real* l = s->lam[k][j];
            real a = qp->r[k][j] - l[0] + l[1];
            for (int i = 0; i < NX; ++i) a += qp->B[k][i][j] * W->pia[k][i];
            UPD1(rs, a); UPD1(rs, qp->rho_l - l[0] - l[2]); UPD1(rs, qp->rho_u - l[1] - l[3]);
            UPD1(ri, -qp->dlu[k][j] + t[2] - t[0]); UPD1(ri, qp->duu[k][j] + t[3] - t[1]);
            for (int i = 0; i < 4; ++i) UPD1(rc, l[i] * t[i]);
        }
        for (int i = 0; i < NX; ++i) {
            UPD1(re, qp->b[k][i]);
            real a = qp->q[k][i] - (k >= 1 ? W->pia[k - 1][i] : W->pia[N][i]);
            if (i == 6 && k >= 1) a += -s->lamd[k][0] + s->lamd[k][1];
            for (int l = 0; l < NX; ++l) a += qp->A[k][l][i] * W->pia[k][l];
            UPD1(rs, a);
        }
        if (k >= 1) {
            UPD1(ri, -qp->dld[k] - s->td[k][0]); UPD1(ri, qp->dud[k] - s->td[k][1]);
            UPD1(rc, s->lamd[k][0] * s->td[k][0]); UPD1(rc, s->lamd[k][1] * s->td[k][1]);
        }
    }
    for (int i = 0; i < NX; ++i) { UPD1(rs, qp->q[N][i] - W->pia[N - 1][i]); UPD1(re, qp->dx0[i]); }
#undef UPD1
    res[0] = rs; res[1] = re; res[2] = ri; res[3] = rc;
}

/* one RTI step; built != 0: the caller has linearised at this iterate already (build_qp) */
static int rti_step(const AdmpcConfig* c, Workspace* W, const double* x0, const double* yref, const double* yref_e,
                    double p, double* xbar, double* ubar, int* iters, int built)
{
    const int N = c->N;
    StageQP* qp = &W->qp;
    if (!built) build_qp(c, W, x0, yref, yref_e, p, xbar, ubar);
    int it = ipm_solve(c, qp, &W->st, &W->wk, &W->F);                       /* H4+H5 */
    if (iters) *iters = it;
    if (it < 0) return ADMPC_STATUS_QP_FAILURE;
    /* H6: expand dx from du through the linearised dynamics, full step */
    for (int i = 0; i < NX; ++i) W->dx[0][i] = qp->dx0[i];
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < NX; ++i) {
            real a = qp->b[k][i];
            for (int l = 0; l < NX; ++l) a += qp->A[k][i][l] * W->dx[k][l];
            for (int l = 0; l < NU; ++l) a += qp->B[k][i][l] * W->st.du[k][l];
            W->dx[k + 1][i] = a;
        }
    int bad = 0;      /* a non-finite step is a QP failure: the iterate is left untouched (acados returns before the update) */
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) { real v = (real)xbar[k * NX + i] + W->dx[k][i]; if (!(R_FABS(v) <= 1e300)) bad = 1; }
    for (int k = 0; k < N; ++k) for (int j = 0; j < NU; ++j) { real v = (real)ubar[k * NU + j] + W->st.du[k][j]; if (!(R_FABS(v) <= 1e300)) bad = 1; }
    if (bad) return ADMPC_STATUS_QP_FAILURE;
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) xbar[k * NX + i] = (double)((real)xbar[k * NX + i] + W->dx[k][i]);
    for (int k = 0; k < N; ++k) for (int j = 0; j < NU; ++j) ubar[k * NU + j] = (double)((real)ubar[k * NU + j] + W->st.du[k][j]);
    return bad ? ADMPC_STATUS_QP_FAILURE : ADMPC_STATUS_SUCCESS;
}

static double eval_cost(const AdmpcConfig* c, const double* yref, const double* yref_e, const double* xbar, const double* ubar)
{
    const int N = c->N;
    real J = 0;
    for (int k = 0; k < N; ++k) {
        for (int i = 0; i < NX; ++i) { real e = (real)xbar[k * NX + i] - (real)yref[k * NY + i]; J += (real)0.5 * c->Ts * c->W[i] * e * e; }
        for (int j = 0; j < NU; ++j) {
            real u = ubar[k * NU + j];
            real e = u - (real)yref[k * NY + NX + j]; J += (real)0.5 * c->Ts * c->W[NX + j] * e * e;
            if (u < c->lbu[j]) J += c->Ts * c->zl * (c->lbu[j] - u);
            if (u > c->ubu[j]) J += c->Ts * c->zu * (u - c->ubu[j]);
        }
    }
    for (int i = 0; i < NX; ++i) { real e = (real)xbar[N * NX + i] - (real)yref_e[i]; J += (real)0.5 * c->We[i] * e * e; }
    return (double)J;
}

/* ------------------------------------------------------------------------------------------ */
/* exported                                                                                   */
/* ------------------------------------------------------------------------------------------ */
void oracle_f(const AdmpcConfig* c, const double* x, const double* u, double p, double* xdot)
{
    real xr[NX], ur[NU], f[NX];
    for (int i = 0; i < NX; ++i) xr[i] = x[i];
    for (int j = 0; j < NU; ++j) ur[j] = u[j];
    model_f(c, xr, ur, (real)p, f);
    for (int i = 0; i < NX; ++i) xdot[i] = (double)f[i];
}

void oracle_jac(const AdmpcConfig* c, const double* x, const double* u, double p, double* Jx, double* Ju)
{
    real xr[NX], ur[NU], jx[NX][NX], ju[NX][NU];
    for (int i = 0; i < NX; ++i) xr[i] = x[i];
    for (int j = 0; j < NU; ++j) ur[j] = u[j];
    model_jac(c, xr, ur, (real)p, jx, ju);
    for (int i = 0; i < NX; ++i) { for (int j = 0; j < NX; ++j) Jx[i * NX + j] = (double)jx[i][j]; for (int j = 0; j < NU; ++j) Ju[i * NU + j] = (double)ju[i][j]; }
}

void oracle_rk4_sens(const AdmpcConfig* c, const double* x, const double* u, double p, double h, double* phi, double* A, double* B)
{
    real xr[NX], ur[NU], ph[NX], a[NX][NX], b[NX][NU];
    for (int i = 0; i < NX; ++i) xr[i] = x[i];
    for (int j = 0; j < NU; ++j) ur[j] = u[j];
    rk4_sens(c, xr, ur, (real)p, (real)h, ph, a, b);
    for (int i = 0; i < NX; ++i) { phi[i] = (double)ph[i]; for (int j = 0; j < NX; ++j) A[i * NX + j] = (double)a[i][j]; for (int j = 0; j < NU; ++j) B[i * NU + j] = (double)b[i][j]; }
}

/* Receding-horizon shift of the iterate (same argument meaning as admpc_shift_batch, host pointers).  Not a reference
 * function: the reference keeps its iterate as it is (acados_solver_sim_car.c:705-731); SURVEY 8f-3 lists it as an option. */
int oracle_shift_batch(const AdmpcConfig* c, int B, double* xbar, double* ubar, const double* p, int rollout)
{
    if (!c || c->N < 2 || c->N > MAXN || B < 0) return ADMPC_EINVAL;
    const int N = c->N;
    for (int b = 0; b < B; ++b) {
        double* xb = xbar + (size_t)b * (N + 1) * NX;
        double* ub = ubar + (size_t)b * N * NU;
        double xn[NX];
        for (int i = 0; i < NX; ++i) xn[i] = xb[N * NX + i];
        if (rollout) {
            real xr[NX], ur[NU], ph[NX], a[NX][NX], bm[NX][NU];
            for (int i = 0; i < NX; ++i) xr[i] = xb[N * NX + i];
            for (int j = 0; j < NU; ++j) ur[j] = ub[(N - 1) * NU + j];
            rk4_sens(c, xr, ur, (real)p[b], (real)c->Ts, ph, a, bm);
            for (int i = 0; i < NX; ++i) xn[i] = (double)ph[i];
        }
        for (int i = 0; i < N * NX; ++i) xb[i] = xb[i + NX];
        for (int i = 0; i < NX; ++i) xb[N * NX + i] = xn[i];
        for (int i = 0; i < (N - 1) * NU; ++i) ub[i] = ub[i + NU];
    }
    return ADMPC_OK;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_solve_batch(const AdmpcConfig* c, int B, const double* x0, const double* yref, const double* yref_e, const double* p,
                       double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, int nthreads)
{
    if (!c || c->N < 2 || c->N > MAXN || B < 0) return ADMPC_EINVAL;
    const int N = c->N;
    const int nsqp = c->sqp_iters > 0 ? c->sqp_iters : 1;
    (void)nthreads;
    int err = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        Workspace* W = (Workspace*)malloc(sizeof(Workspace));
        if (!W) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            err = 1;
        } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
            for (int b = 0; b < B; ++b) {
                double* xb = xbar + (size_t)b * (N + 1) * NX;
                double* ub = ubar + (size_t)b * N * NU;
                const double* yr = yref + (size_t)b * N * NY;
                const double* ye = yref_e + (size_t)b * NX;
                int st = 0, it = 0;
                const int tol_on = nsqp > 1 && c->sqp_tol > 0;      /* reference solver_type "SQP": stop on tolerance, status 2 at the limit */
                int conv = 0;
                for (int s = 0; s < nsqp && st == 0 && !conv; ++s) {
                    if (tol_on) {
                        /* acados' loop: linearise, test the four residuals with the multipliers of the previous QP, solve.  The first
                         * pass has no multipliers to test with (this entry point takes none: a cold solver) and always solves. */
                        build_qp(c, W, x0 + (size_t)b * NX, yr, ye, p[b], xb, ub);
                        if (s > 0) {
                            real res[4];
                            nlp_residuals(W, res);
                            if (res[0] <= c->sqp_tol && res[1] <= c->sqp_tol && res[2] <= c->sqp_tol && res[3] <= c->sqp_tol) { conv = 1; break; }
                        }
                    }
                    st = rti_step(c, W, x0 + (size_t)b * NX, yr, ye, p[b], xb, ub, &it, tol_on);
                    if (st == 0 && tol_on) adjoint_multipliers(c, W, yr, ye, xb);
                }
                if (st == 0 && tol_on && !conv) st = ADMPC_STATUS_MAXITER;
                if (status) status[b] = st;
                if (iters) iters[b] = it;
                if (cost) cost[b] = (st == 0 || st == ADMPC_STATUS_MAXITER) ? eval_cost(c, yr, ye, xb, ub) : INFINITY;
            }
            free(W);
        }
    }
    return err ? ADMPC_ENOMEM : ADMPC_OK;
}

int oracle_qp_debug(const AdmpcConfig* c, const double* x0, const double* yref, const double* yref_e, double p,
                    const double* xbar_in, const double* ubar_in,
                    double* du, double* dx, double* A, double* Bm, double* b,
                    double* lam_u, double* lam_d, double* sl, double* su, int32_t* iters)
{
    if (!c || c->N < 2 || c->N > MAXN) return ADMPC_EINVAL;
    const int N = c->N;
    Workspace* W = (Workspace*)malloc(sizeof(Workspace));
    double* xb = (double*)malloc(sizeof(double) * (N + 1) * NX);
    double* ub = (double*)malloc(sizeof(double) * N * NU);
    if (!W || !xb || !ub) { free(W); free(xb); free(ub); return ADMPC_ENOMEM; }
    memcpy(xb, xbar_in, sizeof(double) * (N + 1) * NX);
    memcpy(ub, ubar_in, sizeof(double) * N * NU);
    int it = 0;
    int st = rti_step(c, W, x0, yref, yref_e, p, xb, ub, &it, 0);
    if (iters) *iters = it;
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j < NU; ++j) {
            du[k * NU + j] = (double)W->st.du[k][j];
            sl[k * NU + j] = (double)W->st.sl[k][j]; su[k * NU + j] = (double)W->st.su[k][j];
            for (int i = 0; i < 4; ++i) lam_u[(k * NU + j) * 4 + i] = (double)W->st.lam[k][j][i];
        }
        lam_d[k * 2 + 0] = k >= 1 ? (double)W->st.lamd[k][0] : 0; lam_d[k * 2 + 1] = k >= 1 ? (double)W->st.lamd[k][1] : 0;
        for (int i = 0; i < NX; ++i) {
            b[k * NX + i] = (double)W->qp.b[k][i];
            for (int j = 0; j < NX; ++j) A[(k * NX + i) * NX + j] = (double)W->qp.A[k][i][j];
            for (int j = 0; j < NU; ++j) Bm[(k * NX + i) * NU + j] = (double)W->qp.B[k][i][j];
        }
    }
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) dx[k * NX + i] = (double)W->dx[k][i];
    free(W); free(xb); free(ub);
    return st;
}

/* acados' four stopping residuals (res_stat, res_eq, res_ineq, res_comp) at an iterate with given multipliers, in the record layout of
 * include/admpc.h (pi [N+1][7], ineq [N][20] = t[10], lam[10]): the checker of admpc_nlp_residuals_batch / admpc_nlp_res_kernel.  One instance. */
int oracle_nlp_residuals(const AdmpcConfig* c, const double* x0, const double* yref, const double* yref_e, double p,
                         const double* xbar, const double* ubar, const double* pi, const double* ineq, double* res)
{
    if (!c || c->N < 2 || c->N > MAXN) return ADMPC_EINVAL;
    const int N = c->N;
    Workspace* W = (Workspace*)malloc(sizeof(Workspace));
    if (!W) return ADMPC_ENOMEM;
    build_qp(c, W, x0, yref, yref_e, p, xbar, ubar);
    IpmState* s = &W->st;
    for (int k = 0; k < N; ++k) {
        const double* t = ineq + (size_t)k * 20; const double* l = t + 10;
        for (int j = 0; j < NU; ++j) {
            s->t[k][j][0] = (real)t[2 * j]; s->t[k][j][1] = (real)t[2 * j + 1]; s->t[k][j][2] = (real)t[6 + 2 * j]; s->t[k][j][3] = (real)t[7 + 2 * j];
            s->lam[k][j][0] = (real)l[2 * j]; s->lam[k][j][1] = (real)l[2 * j + 1]; s->lam[k][j][2] = (real)l[6 + 2 * j]; s->lam[k][j][3] = (real)l[7 + 2 * j];
        }
        s->td[k][0] = (real)t[4]; s->td[k][1] = (real)t[5]; s->lamd[k][0] = (real)l[4]; s->lamd[k][1] = (real)l[5];
    }
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) W->pia[k][i] = (real)pi[(size_t)k * NX + i];
    real r[4];
    nlp_residuals(W, r);
    for (int i = 0; i < 4; ++i) res[i] = (double)r[i];
    free(W);
    return ADMPC_OK;
}

/* scripts/ only: the interior-point trace of one RTI step of one instance: trace [cap][2] = (mu, alpha) per iteration; returns the iteration count */
int oracle_ipm_trace(const AdmpcConfig* c, const double* x0, const double* yref, const double* yref_e, double p,
                     const double* xbar_in, const double* ubar_in, double* trace, int cap)
{
    if (!c || c->N < 2 || c->N > MAXN) return -1;
    const int N = c->N;
    Workspace* W = (Workspace*)malloc(sizeof(Workspace));
    double* xb = (double*)malloc(sizeof(double) * (N + 1) * NX);
    double* ub = (double*)malloc(sizeof(double) * N * NU);
    if (!W || !xb || !ub) { free(W); free(xb); free(ub); return -1; }
    memcpy(xb, xbar_in, sizeof(double) * (N + 1) * NX);
    memcpy(ub, ubar_in, sizeof(double) * N * NU);
    for (int i = 0; i < 2 * cap; ++i) trace[i] = 0;
    g_trace = trace; g_trace_cap = cap;
    int it = 0;
    (void)rti_step(c, W, x0, yref, yref_e, p, xb, ub, &it, 0);
    g_trace = 0; g_trace_cap = 0;
    free(W); free(xb); free(ub);
    return it;
}
