// admpc_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the batched AD-MPC solve engine.
//
// One MPC instance per 64-lane wavefront (workgroup = 1 wave); all per-stage data of the
// instance lives in LDS; the per-constraint interior-point state lives in registers.
//
// Hot path restated (reference = data_driven_mpc/ros_gp_mpc/src/ad_mpc/...):
//   H0/H1  model + ERK4 with forward sensitivities   ad_3d_optimizer.py:280-310, acados ERK
//                                                     (acados_solver_sim_car.c:655-665)
//   H2/H3  Gauss-Newton LS cost, soft/hard bounds     ad_3d_optimizer.py:146-199
//   H4/H5  QP solve: stage-wise Riccati factorisation inside a Mehrotra predictor-corrector
//          primal-dual IPM (reference: full condensing + HPIPM, acados_solver_sim_car.c:145,688-692;
//          same unique minimiser)
//   H6     full step update of the iterate            acados_solver_sim_car.c:647-648,677
//
// Lane roles (N = horizon):
//   shooting      task t=3k+g  -> stage k, sensitivity column group g in {x-cols 2..4, x-cols 5..6, u-cols}
//   u-constraint  set  s=2k+j  -> soft box on input j of stage k (4 inequalities), UPL sets per lane
//   d-constraint  stage k      -> hard box on delta (state 6), stages 1..N-1
//   Riccati       lane l<63    -> matrix entry (l/9,l%9) / (l/7,l%7) of the 7x9 / 9x7 stage products
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/admpc.h"

#define NX ADMPC_NX
#define NU ADMPC_NU
#define NY ADMPC_NY
#define WAVE 64
#define GS 63            // doubles per stage of G = [A|B] (7x9 row-major); odd stride -> conflict-free stage-parallel reads
#define PS 49            // doubles per stage of P (7x7)
#define KS 15            // doubles per stage of K (2x7) padded to odd
#define IPM_FLOOR 1e-40

namespace {

// ---------------------------------------------------------------------------------------------
// wave-wide reductions (xor butterfly: every lane ends with the bitwise identical result)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
    return v;
}
// NaN-propagating max of |v| (fmax would drop NaNs)
__device__ __forceinline__ double absmax_nan(double acc, double v) {
    double a = fabs(v);
    return (a > acc || a != a) ? a : acc;
}
__device__ __forceinline__ double wave_max_nan(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double w = __shfl_xor(v, o, WAVE); v = (w > v || w != w) ? w : v; }
    return v;
}
__device__ __forceinline__ double uniform(double v) {   // make a wave-uniform value provably uniform (SGPR)
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// model: f and the non-zero pattern of its Jacobians (ad_3d_optimizer.py:280-310)
// ---------------------------------------------------------------------------------------------
struct ModelEval {
    double f[NX];
    double j0[3], j1[3];      // rows 0,1 of Jx over (psi, vx, vy)
    double a[3][4];           // rows 3,4,5 of Jx over (vx, vy, psi_dot, delta)
    double bu[3][2];          // rows 3,4,5 of Ju
};

__device__ __forceinline__ void gp_eval(const AdmpcGp& g, double z, double& mu, double& dmu) {
    double m = 0.0, d = 0.0;
    const int n = g.n_points;
    for (int i = 0; i < n; ++i) {
        double dz = z - g.Z[i];
        double k = g.sigma_f * exp(-0.5 * dz * dz * g.inv_l2);
        m += k * g.alpha[i];
        d -= k * dz * g.inv_l2 * g.alpha[i];
    }
    mu = m + g.ymean; dmu = d;
}

__device__ __forceinline__ void model_eval(const AdmpcConfig* __restrict__ c, const double* x, const double* u, double p, ModelEval& e)
{
    const double psi = x[2], vx = x[3], vy = x[4], r = x[5], dl = x[6];
    const double m = c->mass, LF = c->L_F, LR = c->L_R, Iz = c->Iz, Cf = c->Cf, Cr = c->Cr;
    const double L = LR + LF;
    double sp, cp, sd, cd;
    sincos(psi, &sp, &cp);
    sincos(dl, &sd, &cd);
    e.f[0] = vx * cp - vy * sp;
    e.f[1] = vx * sp + vy * cp;
    e.f[2] = r;
    e.j0[0] = -vx * sp - vy * cp; e.j0[1] = cp; e.j0[2] = -sp;
    e.j1[0] = vx * cp - vy * sp;  e.j1[1] = sp; e.j1[2] = cp;
    const double v = vx + 1e-99;
    const double iv = 1.0 / v;
    const double Ffy = 2 * Cf * (dl - (vy + LF * r) * iv);
    const double Fry = 2 * Cr * (LR * r - vy) * iv;
    const double im = 1.0 / m, iIz = 1.0 / Iz;
    const double kk = u[1] * vx + dl * u[0];
    const double dyn3 = u[0] - im * Ffy * sd + vy * r;
    const double dyn4 = im * (Fry + Ffy * cd) - vx * r;
    const double dyn5 = iIz * (LF * Ffy * cd - LR * Fry);
    const double q = 1.0 - p;
    e.f[3] = p * dyn3 + q * u[0];
    e.f[4] = p * dyn4 + q * (kk * LR / L);
    e.f[5] = p * dyn5 + q * (kk / L);
    e.f[6] = u[1];
    const double gF[4] = { 2 * Cf * (vy + LF * r) * iv * iv, -2 * Cf * iv, -2 * Cf * LF * iv, 2 * Cf };
    const double gR[4] = { -Fry * iv, -2 * Cr * iv, 2 * Cr * LR * iv, 0.0 };
    double d3[4], d4[4], d5[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        d3[i] = -gF[i] * sd * im;
        d4[i] = (gR[i] + gF[i] * cd) * im;
        d5[i] = (LF * gF[i] * cd - LR * gR[i]) * iIz;
    }
    d3[1] += r;  d3[2] += vy;  d3[3] += -Ffy * cd * im;
    d4[0] += -r; d4[2] += -vx; d4[3] += -Ffy * sd * im;
    d5[3] += -LF * Ffy * sd * iIz;
    const double k4[4] = { u[1] * LR / L, 0.0, 0.0, u[0] * LR / L };
    const double k5[4] = { u[1] / L, 0.0, 0.0, u[0] / L };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e.a[0][i] = p * d3[i];
        e.a[1][i] = p * d4[i] + q * k4[i];
        e.a[2][i] = p * d5[i] + q * k5[i];
    }
    e.bu[0][0] = 1.0;              e.bu[0][1] = 0.0;
    e.bu[1][0] = q * dl * LR / L;  e.bu[1][1] = q * vx * LR / L;
    e.bu[2][0] = q * dl / L;       e.bu[2][1] = q * vx / L;
    const int ngp = c->n_gp;
    for (int g = 0; g < ngp; ++g) {          // residual GPs: out in {3,4,5}, feat in {3..8} (validated on the host)
        const AdmpcGp& gp = c->gp[g];
        const int feat = gp.feat - 3, out = gp.out - 3;          // feat: 0..3 -> (vx,vy,r,delta), 4..5 -> (u0,u1)
        // static indexing only: runtime-indexed private arrays would live in scratch memory
        const double z = feat == 0 ? vx : feat == 1 ? vy : feat == 2 ? r : feat == 3 ? dl : feat == 4 ? u[0] : u[1];
        double mu, dmu;
        gp_eval(gp, z, mu, dmu);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const bool so = out == o;
            e.f[3 + o] += so ? mu : 0.0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) e.a[o][cc] += (so && feat == cc) ? dmu : 0.0;
            e.bu[o][0] += (so && feat == 4) ? dmu : 0.0;
            e.bu[o][1] += (so && feat == 5) ? dmu : 0.0;
        }
    }
}

// d(column)/dt = Jx * s (+ Ju column for an input column)
__device__ __forceinline__ void sens_rhs(const ModelEval& e, const double* s, int ucol, double* d)
{
    d[0] = e.j0[0] * s[2] + e.j0[1] * s[3] + e.j0[2] * s[4];
    d[1] = e.j1[0] * s[2] + e.j1[1] * s[3] + e.j1[2] * s[4];
    d[2] = s[5];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double a = e.a[r][0] * s[3] + e.a[r][1] * s[4] + e.a[r][2] * s[5] + e.a[r][3] * s[6];
        a += ucol == 0 ? e.bu[r][0] : (ucol == 1 ? e.bu[r][1] : 0.0);
        d[3 + r] = a;
    }
    d[6] = ucol == 1 ? 1.0 : 0.0;
}

// One ERK4 step of length h for the state and for the NC sensitivity columns of group g:
//   g=0: x-columns 2,3,4   g=1: x-columns 5,6   g=2: u-columns 0,1
// Results: phi[7] (all groups), col[c][7] = column c of the group of A (g<2) or B (g=2).
__device__ __forceinline__ void rk4_group(const AdmpcConfig* __restrict__ c, const double* x, const double* u, double p, double h,
                                          int g, double* phi, double col[3][NX])
{
    const int xcol0 = g == 0 ? 2 : 5;
    double kx[NX], accx[NX];
    double kS[3][NX], accS[3][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { kx[i] = 0.0; accx[i] = 0.0; }
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int i = 0; i < NX; ++i) { kS[cc][i] = 0.0; accS[cc][i] = 0.0; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double cs = (s == 0) ? 0.0 : (s == 3 ? 1.0 : 0.5);
        const double ws = (s == 0 || s == 3) ? (1.0 / 6.0) : (2.0 / 6.0);
        double X[NX], S[3][NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) X[i] = x[i] + cs * h * kx[i];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double id = (g < 2 && i == xcol0 + cc) ? 1.0 : 0.0;
                S[cc][i] = id + cs * h * kS[cc][i];
            }
        ModelEval e;
        model_eval(c, X, u, p, e);
#pragma unroll
        for (int i = 0; i < NX; ++i) { kx[i] = e.f[i]; accx[i] += ws * e.f[i]; }
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            sens_rhs(e, S[cc], g == 2 ? cc : -1, kS[cc]);
#pragma unroll
            for (int i = 0; i < NX; ++i) accS[cc][i] += ws * kS[cc][i];
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) phi[i] = x[i] + h * accx[i];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double id = (g < 2 && i == xcol0 + cc) ? 1.0 : 0.0;
            col[cc][i] = id + h * accS[cc][i];
        }
}

// ---------------------------------------------------------------------------------------------
// LDS map of one instance (offsets in doubles)
// ---------------------------------------------------------------------------------------------
struct Lds {
    double *G, *P, *K, *Li, *bl, *q, *r, *dx, *pi, *gx, *gu, *pv, *kff, *ddx, *ddu, *Rt, *Qt, *lamd, *Ms, *Hux, *Huu;
    double *xs, *us, *yr, *ye, *x0;
};
__host__ __device__ inline int lds_doubles(int N) {
    return N * GS + (N + 1) * PS + N * KS + N * 3 + N * 7 + (N + 1) * 7 + N * 2 + (N + 1) * 7 + N * 7 + (N + 1) * 7 + N * 2 +
           (N + 1) * 7 + N * 2 + (N + 1) * 7 + N * 2 + N * 2 + N + N * 2 + 64 + 14 + 4 +
           (N + 1) * 7 + N * 2 + N * 9 + 7 + 7;
}
__device__ __forceinline__ void lds_carve(double* base, int N, Lds& L) {
    double* p = base;
    L.G = p; p += N * GS;        L.P = p; p += (N + 1) * PS;  L.K = p; p += N * KS;       L.Li = p; p += N * 3;
    L.bl = p; p += N * 7;        L.q = p; p += (N + 1) * 7;   L.r = p; p += N * 2;        L.dx = p; p += (N + 1) * 7;
    L.pi = p; p += N * 7;        L.gx = p; p += (N + 1) * 7;  L.gu = p; p += N * 2;       L.pv = p; p += (N + 1) * 7;
    L.kff = p; p += N * 2;       L.ddx = p; p += (N + 1) * 7; L.ddu = p; p += N * 2;      L.Rt = p; p += N * 2;
    L.Qt = p; p += N;            L.lamd = p; p += N * 2;      L.Ms = p; p += 64;          L.Hux = p; p += 14;
    L.Huu = p; p += 4;
    L.xs = p; p += (N + 1) * 7;  L.us = p; p += N * 2;        L.yr = p; p += N * 9;       L.ye = p; p += 7;  L.x0 = p; p += 7;
}

#define WSYNC() __syncthreads()

// ---------------------------------------------------------------------------------------------
// Riccati: matrices (backward), vectors (backward), roll-out (forward)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void riccati_factor(const Lds& L, int N, int lane, const double* Qd, const double* Qe)
{
    if (lane < PS) { int i = lane / 7, j = lane % 7; L.P[N * PS + lane] = (i == j) ? Qe[i] : 0.0; }
    WSYNC();
    const int i1 = lane / 9, j1 = lane % 9;         // L1: M = P+ * G  (7x9)
    const int a2 = lane / 7, j2 = lane % 7;         // L2: H[a][j] = G[:,a]' M[:,j]  (9x7), L3: P entry (a2,j2) for lane<49
    const int ua = (lane >> 1) & 1, ub = lane & 1;  // Huu entry computed redundantly
    for (int k = N - 1; k >= 0; --k) {
        const double* G = L.G + k * GS;
        const double* Pn = L.P + (k + 1) * PS;
        if (lane < 63) {
            double m = 0.0;
#pragma unroll
            for (int l = 0; l < NX; ++l) m += Pn[i1 * 7 + l] * G[l * 9 + j1];
            L.Ms[lane] = m;
        }
        WSYNC();
        double hxx = 0.0;
        if (lane < 63) {
            double hv = 0.0;
#pragma unroll
            for (int l = 0; l < NX; ++l) hv += G[l * 9 + a2] * L.Ms[l * 9 + j2];
            if (a2 == j2) hv += (a2 == 6 && k >= 1) ? L.Qt[k] : Qd[a2];
            if (a2 >= NX) L.Hux[(a2 - NX) * 7 + j2] = hv; else hxx = hv;
        }
        {
            double hu = 0.0;
#pragma unroll
            for (int l = 0; l < NX; ++l) hu += G[l * 9 + 7 + ua] * L.Ms[l * 9 + 7 + ub];
            if (ua == ub) hu += L.Rt[k * 2 + ua];
            if (lane < 4) L.Huu[lane] = hu;
        }
        WSYNC();
        {
            const double h00 = L.Huu[0], h01 = L.Huu[1], h11 = L.Huu[3];
            const double det = h00 * h11 - h01 * h01;
            const double idet = 1.0 / det;
            const double i00 = h11 * idet, i01 = -h01 * idet, i11 = h00 * idet;
            if (lane < PS) {
                const double h0i = L.Hux[a2], h1i = L.Hux[7 + a2], h0j = L.Hux[j2], h1j = L.Hux[7 + j2];
                const double K0 = -(i00 * h0j + i01 * h1j), K1 = -(i01 * h0j + i11 * h1j);
                L.P[k * PS + lane] = hxx + h0i * K0 + h1i * K1;
                if (a2 == 0) { L.K[k * KS + j2] = K0; L.K[k * KS + 7 + j2] = K1; }
            }
            if (lane == 0) { L.Li[k * 3 + 0] = i00; L.Li[k * 3 + 1] = i01; L.Li[k * 3 + 2] = i11; }
        }
        WSYNC();
    }
}

// backward sweep of the gradient (pv) and feed-forward (kff); the dynamics residual of the Newton
// system is identically zero because every IPM iterate satisfies the linearised dynamics.
__device__ __forceinline__ void riccati_backward(const Lds& L, int N, int lane)
{
    if (lane < NX) L.pv[N * 7 + lane] = L.gx[N * 7 + lane];
    WSYNC();
    for (int k = N - 1; k >= 0; --k) {
        const double* G = L.G + k * GS;
        const double* pn = L.pv + (k + 1) * 7;
        double h = 0.0;
        if (lane < NY) {
            h = lane < NX ? L.gx[k * 7 + lane] : L.gu[k * 2 + lane - NX];
#pragma unroll
            for (int l = 0; l < NX; ++l) h += G[l * 9 + lane] * pn[l];
        }
        const double hu0 = __shfl(h, 7, WAVE), hu1 = __shfl(h, 8, WAVE);
        if (lane < NX) L.pv[k * 7 + lane] = h + L.K[k * KS + lane] * hu0 + L.K[k * KS + 7 + lane] * hu1;
        if (lane == 0) {
            const double i00 = L.Li[k * 3], i01 = L.Li[k * 3 + 1], i11 = L.Li[k * 3 + 2];
            L.kff[k * 2 + 0] = -(i00 * hu0 + i01 * hu1);
            L.kff[k * 2 + 1] = -(i01 * hu0 + i11 * hu1);
        }
        WSYNC();
    }
}

__device__ __forceinline__ void riccati_forward(const Lds& L, int N, int lane)
{
    if (lane < NX) L.ddx[lane] = 0.0;
    WSYNC();
    for (int k = 0; k < N; ++k) {
        const double* G = L.G + k * GS;
        const double* xk = L.ddx + k * 7;
        double du = 0.0;
        if (lane < NU) {
            du = L.kff[k * 2 + lane];
#pragma unroll
            for (int l = 0; l < NX; ++l) du += L.K[k * KS + lane * 7 + l] * xk[l];
            L.ddu[k * 2 + lane] = du;
        }
        const double du0 = __shfl(du, 0, WAVE), du1 = __shfl(du, 1, WAVE);
        if (lane < NX) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < NX; ++l) a += G[lane * 9 + l] * xk[l];
            a += G[lane * 9 + 7] * du0 + G[lane * 9 + 8] * du1;
            L.ddx[(k + 1) * 7 + lane] = a;
        }
        WSYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// per-lane interior point state
// ---------------------------------------------------------------------------------------------
struct USet {      // soft box on one input: 0 lower, 1 upper, 2 sl>=0, 3 su>=0
    double t[4], lam[4], du, sl, su, dl, duu, r;      // dl = lbu-ubar, duu = ubu-ubar, r = cost gradient
    double rc[4], rd[4], rsl, rsu, ru, e1, e2, dt[4], dlam[4], dsl, dsu;
};
struct DSet {      // hard box on delta of one stage
    double t[2], lam[2], dl, du;
    double rc[2], rd[2], dt[2], dlam[2];
};

template <int UPL>
__device__ __forceinline__ int ipm_solve(const AdmpcConfig* __restrict__ c, const Lds& L, int N, int lane,
                                         USet (&U)[UPL], DSet& D, const double* Qd, const double* Rd, const double* Qe,
                                         double rho_l, double rho_u, bool& failed)
{
    const double thr = c->ipm_thr0, mu0 = c->ipm_mu0;
    const double tol_comp = c->ipm_tol_comp, tol_res = c->ipm_tol_res, tol_step = c->ipm_tol_step;
    const int itmax = c->ipm_iter_max;
    const double inv_nineq = 1.0 / (double)(8 * N + 2 * (N - 1));
    const int nu_sets = 2 * N;
    const bool dact = lane >= 1 && lane < N;            // this lane owns the delta bounds of stage `lane`
    // ---- cold start: zero input step, states rolled out through the linearised dynamics, slacks at thr
#pragma unroll
    for (int m = 0; m < UPL; ++m) {
        USet& s = U[m];
        s.du = 0.0; s.sl = thr; s.su = thr;
        const double r0[4] = { thr - s.dl, thr + s.duu, thr, thr };
#pragma unroll
        for (int i = 0; i < 4; ++i) { s.t[i] = r0[i] > thr ? r0[i] : thr; s.lam[i] = mu0 / s.t[i]; }
    }
    // dx[0] = x0 - xbar0 ; dx[k+1] = A dx[k] + b   (du = 0)
    if (lane < NX) L.dx[lane] = L.x0[lane] - L.xs[lane];
    for (int i = lane; i < N * 7; i += WAVE) L.pi[i] = 0.0;
    WSYNC();
    for (int k = 0; k < N; ++k) {
        if (lane < NX) {
            double a = L.bl[k * 7 + lane];
#pragma unroll
            for (int l = 0; l < NX; ++l) a += L.G[k * GS + lane * 9 + l] * L.dx[k * 7 + l];
            L.dx[(k + 1) * 7 + lane] = a;
        }
        WSYNC();
    }
    {
        D.t[0] = D.t[1] = 1.0; D.lam[0] = D.lam[1] = 0.0;
        if (dact) {
            const double x6 = L.dx[lane * 7 + 6];
            const double r0[2] = { x6 - D.dl, D.du - x6 };
#pragma unroll
            for (int i = 0; i < 2; ++i) { D.t[i] = r0[i] > thr ? r0[i] : thr; D.lam[i] = mu0 / D.t[i]; }
        }
    }
    failed = false;
    double rmax_prev = 0.0, step = 1e300;
    int it = 0;
    for (; it < itmax; ++it) {
        // ---- complementarity products, mu
        double musum = 0.0, cmax = 0.0, rmax = 0.0;
#pragma unroll
        for (int m = 0; m < UPL; ++m) {
            USet& s = U[m];
            const bool act = lane + WAVE * m < nu_sets;
#pragma unroll
            for (int i = 0; i < 4; ++i) { s.rc[i] = s.t[i] * s.lam[i]; if (act) { musum += s.rc[i]; cmax = fmax(cmax, s.rc[i]); } }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) { D.rc[i] = D.t[i] * D.lam[i]; if (dact) { musum += D.rc[i]; cmax = fmax(cmax, D.rc[i]); } }
        if (dact) { L.lamd[lane * 2] = D.lam[0]; L.lamd[lane * 2 + 1] = D.lam[1]; }
        WSYNC();
        // ---- linear residuals: input sets
#pragma unroll
        for (int m = 0; m < UPL; ++m) {
            USet& s = U[m];
            const int sid = lane + WAVE * m;
            if (sid < nu_sets) {
                const int k = sid >> 1, j = sid & 1;
                double a = Rd[j] * s.du + s.r - s.lam[0] + s.lam[1];
#pragma unroll
                for (int l = 0; l < NX; ++l) a += L.G[k * GS + l * 9 + 7 + j] * L.pi[k * 7 + l];
                s.ru = a;
                s.rsl = rho_l - s.lam[0] - s.lam[2];
                s.rsu = rho_u - s.lam[1] - s.lam[3];
                s.rd[0] = s.du + s.sl - s.dl - s.t[0];
                s.rd[1] = -s.du + s.su + s.duu - s.t[1];
                s.rd[2] = s.sl - s.t[2];
                s.rd[3] = s.su - s.t[3];
                rmax = absmax_nan(rmax, s.ru); rmax = absmax_nan(rmax, s.rsl); rmax = absmax_nan(rmax, s.rsu);
#pragma unroll
                for (int i = 0; i < 4; ++i) rmax = absmax_nan(rmax, s.rd[i]);
            }
        }
        // ---- state stationarity rows (k,i), k = 1..N
        for (int tsk = lane; tsk < N * 7; tsk += WAVE) {
            const int k = tsk / 7 + 1, i = tsk % 7;
            double a;
            if (k < N) {
                a = Qd[i] * L.dx[k * 7 + i] + L.q[k * 7 + i] - L.pi[(k - 1) * 7 + i];
#pragma unroll
                for (int l = 0; l < NX; ++l) a += L.G[k * GS + l * 9 + i] * L.pi[k * 7 + l];
                if (i == 6) a += -L.lamd[k * 2] + L.lamd[k * 2 + 1];
            } else {
                a = Qe[i] * L.dx[N * 7 + i] + L.q[N * 7 + i] - L.pi[(N - 1) * 7 + i];
            }
            L.gx[k * 7 + i] = a;
            rmax = absmax_nan(rmax, a);
        }
        if (lane < NX) L.gx[lane] = 0.0;
        if (dact) {
            const double x6 = L.dx[lane * 7 + 6];
            D.rd[0] = x6 - D.dl - D.t[0];
            D.rd[1] = D.du - x6 - D.t[1];
            rmax = absmax_nan(rmax, D.rd[0]); rmax = absmax_nan(rmax, D.rd[1]);
        }
        const double mu = uniform(wave_sum(musum)) * inv_nineq;
        cmax = uniform(wave_max(cmax));
        rmax = uniform(wave_max_nan(rmax));
        if (!(mu == mu) || !(rmax == rmax)) { failed = true; break; }
        if (cmax <= tol_comp && step <= tol_step && (rmax <= tol_res || (it > 0 && rmax > 0.1 * rmax_prev))) break;
        rmax_prev = rmax;
        WSYNC();

        double mu_aff = 0.0, sigma = 0.0, rx6 = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {            // 0: predictor, 1: corrector
            // ---- eliminate slacks / multipliers -> barrier-augmented diagonals and gradients
#pragma unroll
            for (int m = 0; m < UPL; ++m) {
                USet& s = U[m];
                const int sid = lane + WAVE * m;
                if (sid < nu_sets) {
                    const double G0 = s.lam[0] / s.t[0], G1 = s.lam[1] / s.t[1], G2 = s.lam[2] / s.t[2], G3 = s.lam[3] / s.t[3];
                    const double c0 = s.rc[0] / s.t[0], c1 = s.rc[1] / s.t[1], c2 = s.rc[2] / s.t[2], c3 = s.rc[3] / s.t[3];
                    s.e1 = s.rsl + c0 + c2 + G0 * s.rd[0] + G2 * s.rd[2];
                    s.e2 = s.rsu + c1 + c3 + G1 * s.rd[1] + G3 * s.rd[3];
                    const double etal = c0 + G0 * s.rd[0] - G0 * s.e1 / (G0 + G2);
                    const double etau = -c1 - G1 * s.rd[1] + G1 * s.e2 / (G1 + G3);
                    if (pass == 0) L.Rt[sid] = Rd[sid & 1] + G0 * G2 / (G0 + G2) + G1 * G3 / (G1 + G3);
                    L.gu[sid] = s.ru + etal + etau;
                }
            }
            if (dact) {
                const double G5 = D.lam[0] / D.t[0], G6 = D.lam[1] / D.t[1];
                if (pass == 0) L.Qt[lane] = Qd[6] + G5 + G6;
                const double ex = (D.rc[0] / D.t[0] + G5 * D.rd[0]) - (D.rc[1] / D.t[1] + G6 * D.rd[1]);
                if (pass == 0) rx6 = L.gx[lane * 7 + 6];
                L.gx[lane * 7 + 6] = rx6 + ex;
            }
            WSYNC();
            if (pass == 0) riccati_factor(L, N, lane, Qd, Qe);
            riccati_backward(L, N, lane);
            riccati_forward(L, N, lane);
            // ---- recover slack / t / lam steps, step length
            double amax = 1.0;
#pragma unroll
            for (int m = 0; m < UPL; ++m) {
                USet& s = U[m];
                const int sid = lane + WAVE * m;
                if (sid < nu_sets) {
                    const double G[4] = { s.lam[0] / s.t[0], s.lam[1] / s.t[1], s.lam[2] / s.t[2], s.lam[3] / s.t[3] };
                    const double u = L.ddu[sid];
                    s.dsl = -(s.e1 + G[0] * u) / (G[0] + G[2]);
                    s.dsu = -(s.e2 - G[1] * u) / (G[1] + G[3]);
                    s.dt[0] = u + s.dsl + s.rd[0]; s.dt[1] = -u + s.dsu + s.rd[1]; s.dt[2] = s.dsl + s.rd[2]; s.dt[3] = s.dsu + s.rd[3];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        s.dlam[i] = -s.rc[i] / s.t[i] - G[i] * s.dt[i];
                        if (s.dt[i] < 0.0) amax = fmin(amax, -s.t[i] / s.dt[i]);
                        if (s.dlam[i] < 0.0) amax = fmin(amax, -s.lam[i] / s.dlam[i]);
                    }
                }
            }
            if (dact) {
                const double G5 = D.lam[0] / D.t[0], G6 = D.lam[1] / D.t[1];
                const double x6 = L.ddx[lane * 7 + 6];
                D.dt[0] = x6 + D.rd[0];  D.dlam[0] = -D.rc[0] / D.t[0] - G5 * D.dt[0];
                D.dt[1] = -x6 + D.rd[1]; D.dlam[1] = -D.rc[1] / D.t[1] - G6 * D.dt[1];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (D.dt[i] < 0.0) amax = fmin(amax, -D.t[i] / D.dt[i]);
                    if (D.dlam[i] < 0.0) amax = fmin(amax, -D.lam[i] / D.dlam[i]);
                }
            }
            amax = uniform(wave_min(amax));
            if (pass == 0) {
                // affine step: mu_aff, sigma, corrector right-hand side
                double s_aff = 0.0;
#pragma unroll
                for (int m = 0; m < UPL; ++m) {
                    USet& s = U[m];
                    if (lane + WAVE * m < nu_sets)
#pragma unroll
                        for (int i = 0; i < 4; ++i) s_aff += (s.t[i] + amax * s.dt[i]) * (s.lam[i] + amax * s.dlam[i]);
                }
                if (dact)
#pragma unroll
                    for (int i = 0; i < 2; ++i) s_aff += (D.t[i] + amax * D.dt[i]) * (D.lam[i] + amax * D.dlam[i]);
                mu_aff = uniform(wave_sum(s_aff)) * inv_nineq;
                sigma = mu_aff / mu; sigma = sigma * sigma * sigma;
                const double smu = sigma * mu;
#pragma unroll
                for (int m = 0; m < UPL; ++m) {
                    USet& s = U[m];
#pragma unroll
                    for (int i = 0; i < 4; ++i) s.rc[i] = s.t[i] * s.lam[i] + s.dt[i] * s.dlam[i] - smu;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) D.rc[i] = D.t[i] * D.lam[i] + D.dt[i] * D.dlam[i] - smu;
            } else {
                double tau = 1.0 - mu_aff; tau = fmax(tau, 0.995); tau = fmin(tau, 0.999999);
                double alpha = fmin(tau * amax, 1.0);
                double stp = 0.0;
#pragma unroll
                for (int m = 0; m < UPL; ++m) {
                    USet& s = U[m];
                    const int sid = lane + WAVE * m;
                    if (sid < nu_sets) {
                        const double u = L.ddu[sid];
                        stp = fmax(stp, fabs(alpha * u));
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            s.t[i] = fmax(s.t[i] + alpha * s.dt[i], IPM_FLOOR);
                            s.lam[i] = fmax(s.lam[i] + alpha * s.dlam[i], IPM_FLOOR);
                        }
                        s.du += alpha * u; s.sl += alpha * s.dsl; s.su += alpha * s.dsu;
                    }
                }
                if (dact)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        D.t[i] = fmax(D.t[i] + alpha * D.dt[i], IPM_FLOOR);
                        D.lam[i] = fmax(D.lam[i] + alpha * D.dlam[i], IPM_FLOOR);
                    }
                // dx += alpha ddx ; pi += alpha (P ddx + pv)
                for (int tsk = lane; tsk < N * 7; tsk += WAVE) {
                    const int k = tsk / 7, i = tsk % 7;       // pi[k][i] and dx[k+1][i]
                    double dp = L.pv[(k + 1) * 7 + i];
#pragma unroll
                    for (int l = 0; l < NX; ++l) dp += L.P[(k + 1) * PS + i * 7 + l] * L.ddx[(k + 1) * 7 + l];
                    L.pi[k * 7 + i] += alpha * dp;
                    L.dx[(k + 1) * 7 + i] += alpha * L.ddx[(k + 1) * 7 + i];
                }
                step = uniform(wave_max(stp));
                WSYNC();
            }
        }
    }
    return it;
}

// ---------------------------------------------------------------------------------------------
// the fused solve kernel: one instance per wave, grid-stride over instances
// ---------------------------------------------------------------------------------------------
template <int UPL>
__global__ __launch_bounds__(WAVE) void admpc_solve_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                           const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                           const double* __restrict__ yrefeg, const double* __restrict__ pg,
                                                           double* __restrict__ xbarg, double* __restrict__ ubarg,
                                                           double* __restrict__ costg, int32_t* __restrict__ statusg,
                                                           int32_t* __restrict__ itersg)
{
    extern __shared__ double lds_raw[];
    const int lane = threadIdx.x;
    const int N = cfg->N;
    Lds L;
    lds_carve(lds_raw, N, L);
    const double Ts = cfg->Ts;
    double Qd[NX], Qe[NX], Rd[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) { Qd[i] = Ts * cfg->W[i]; Qe[i] = cfg->We[i]; }
#pragma unroll
    for (int j = 0; j < NU; ++j) Rd[j] = Ts * cfg->W[NX + j];
    const double rho_l = Ts * cfg->zl, rho_u = Ts * cfg->zu;
    const int nsqp = cfg->sqp_iters > 0 ? cfg->sqp_iters : 1;

    for (int inst = blockIdx.x; inst < B; inst += gridDim.x) {
        // ---- stage the instance record in LDS (coalesced 8 B / lane)
        {
            const double* gx = xbarg + (size_t)inst * (N + 1) * NX;
            const double* gu = ubarg + (size_t)inst * N * NU;
            const double* gy = yrefg + (size_t)inst * N * NY;
            for (int i = lane; i < (N + 1) * NX; i += WAVE) L.xs[i] = gx[i];
            for (int i = lane; i < N * NU; i += WAVE) L.us[i] = gu[i];
            for (int i = lane; i < N * NY; i += WAVE) L.yr[i] = gy[i];
            if (lane < NX) { L.ye[lane] = yrefeg[(size_t)inst * NX + lane]; L.x0[lane] = x0g[(size_t)inst * NX + lane]; }
        }
        const double p = pg[inst];
        WSYNC();
        int status = ADMPC_STATUS_SUCCESS, iters = 0;
        for (int sq = 0; sq < nsqp && status == ADMPC_STATUS_SUCCESS; ++sq) {
            // ---- H1: shooting, task (k,g)
            for (int tsk = lane; tsk < 3 * N; tsk += WAVE) {
                const int k = tsk / 3, g = tsk % 3;
                double x[NX], u[NU], phi[NX], col[3][NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = L.xs[k * 7 + i];
                u[0] = L.us[k * 2]; u[1] = L.us[k * 2 + 1];
                rk4_group(cfg, x, u, p, Ts, g, phi, col);
                double* G = L.G + k * GS;
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        G[i * 9 + 0] = i == 0 ? 1.0 : 0.0; G[i * 9 + 1] = i == 1 ? 1.0 : 0.0;
                        G[i * 9 + 2] = col[0][i]; G[i * 9 + 3] = col[1][i]; G[i * 9 + 4] = col[2][i];
                        L.bl[k * 7 + i] = phi[i] - L.xs[(k + 1) * 7 + i];
                    }
                } else if (g == 1) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { G[i * 9 + 5] = col[0][i]; G[i * 9 + 6] = col[1][i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { G[i * 9 + 7] = col[0][i]; G[i * 9 + 8] = col[1][i]; }
                }
            }
            // ---- H2: gradients of the Gauss-Newton model
            for (int i = lane; i < (N + 1) * NX; i += WAVE) {
                const int k = i / 7, ii = i % 7;
                L.q[i] = k < N ? Qd[ii] * (L.xs[i] - L.yr[k * 9 + ii]) : Qe[ii] * (L.xs[i] - L.ye[ii]);
            }
            // ---- H3: bounds in step form
            USet U[UPL];
            DSet D;
#pragma unroll
            for (int m = 0; m < UPL; ++m) {
                const int sid = lane + WAVE * m;
                const int sc = sid < 2 * N ? sid : 0;
                const int k = sc >> 1, j = sc & 1;
                const double ub = L.us[sc];
                U[m].dl = cfg->lbu[j] - ub; U[m].duu = cfg->ubu[j] - ub;
                U[m].r = Rd[j] * (ub - L.yr[k * 9 + 7 + j]);
            }
            {
                const int k = lane < N ? lane : 0;
                D.dl = cfg->lbx_delta - L.xs[k * 7 + 6]; D.du = cfg->ubx_delta - L.xs[k * 7 + 6];
            }
            WSYNC();
            // ---- H4/H5: QP
            bool failed;
            iters = ipm_solve<UPL>(cfg, L, N, lane, U, D, Qd, Rd, Qe, rho_l, rho_u, failed);
            // ---- H6: full step
            bool bad = failed;        // a non-finite step is a QP failure: leave the iterate untouched
            for (int i = lane; i < (N + 1) * NX; i += WAVE) { double v = L.xs[i] + L.dx[i]; if (!(fabs(v) <= 1e300)) bad = true; }
#pragma unroll
            for (int m = 0; m < UPL; ++m) {
                const int sid = lane + WAVE * m;
                if (sid < 2 * N) { double v = L.us[sid] + U[m].du; if (!(fabs(v) <= 1e300)) bad = true; }
            }
            if (__any(bad)) {
                status = ADMPC_STATUS_QP_FAILURE;
            } else {
                for (int i = lane; i < (N + 1) * NX; i += WAVE) L.xs[i] += L.dx[i];
#pragma unroll
                for (int m = 0; m < UPL; ++m) {
                    const int sid = lane + WAVE * m;
                    if (sid < 2 * N) L.us[sid] += U[m].du;
                }
            }
            WSYNC();
        }
        // ---- write back, cost
        {
            double* gx = xbarg + (size_t)inst * (N + 1) * NX;
            double* gu = ubarg + (size_t)inst * N * NU;
            for (int i = lane; i < (N + 1) * NX; i += WAVE) gx[i] = L.xs[i];
            for (int i = lane; i < N * NU; i += WAVE) gu[i] = L.us[i];
            double J = 0.0;
            for (int i = lane; i < (N + 1) * NX; i += WAVE) {
                const int k = i / 7, ii = i % 7;
                if (k < N) { double e = L.xs[i] - L.yr[k * 9 + ii]; J += 0.5 * Ts * cfg->W[ii] * e * e; }
                else { double e = L.xs[i] - L.ye[ii]; J += 0.5 * cfg->We[ii] * e * e; }
            }
            for (int i = lane; i < N * NU; i += WAVE) {
                const int k = i >> 1, j = i & 1;
                const double u = L.us[i];
                double e = u - L.yr[k * 9 + 7 + j]; J += 0.5 * Ts * cfg->W[NX + j] * e * e;
                if (u < cfg->lbu[j]) J += Ts * cfg->zl * (cfg->lbu[j] - u);
                if (u > cfg->ubu[j]) J += Ts * cfg->zu * (u - cfg->ubu[j]);
            }
            J = wave_sum(J);
            if (lane == 0) {
                if (costg) costg[inst] = status == ADMPC_STATUS_SUCCESS ? J : INFINITY;
                if (statusg) statusg[inst] = status;
                if (itersg) itersg[inst] = iters;
            }
        }
        WSYNC();
    }
}

// shooting only: phi, A, B to global memory (parity tests of H0/H1)
__global__ __launch_bounds__(WAVE) void admpc_shoot_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                           const double* __restrict__ xbarg, const double* __restrict__ ubarg,
                                                           const double* __restrict__ pg,
                                                           double* __restrict__ phig, double* __restrict__ Ag, double* __restrict__ Bg)
{
    const int N = cfg->N;
    const long total = (long)B * N * 3;
    for (long tsk = (long)blockIdx.x * WAVE + threadIdx.x; tsk < total; tsk += (long)gridDim.x * WAVE) {
        const long sk = tsk / 3; const int g = (int)(tsk % 3);
        const long inst = sk / N; const int k = (int)(sk % N);
        double x[NX], u[NU], phi[NX], col[3][NX];
        for (int i = 0; i < NX; ++i) x[i] = xbarg[(inst * (N + 1) + k) * NX + i];
        u[0] = ubarg[(inst * N + k) * NU]; u[1] = ubarg[(inst * N + k) * NU + 1];
        rk4_group(cfg, x, u, pg[inst], cfg->Ts, g, phi, col);
        double* A = Ag + sk * NX * NX; double* Bm = Bg + sk * NX * NU;
        if (g == 0) {
            for (int i = 0; i < NX; ++i) {
                phig[sk * NX + i] = phi[i];
                A[i * 7 + 0] = i == 0 ? 1.0 : 0.0; A[i * 7 + 1] = i == 1 ? 1.0 : 0.0;
                A[i * 7 + 2] = col[0][i]; A[i * 7 + 3] = col[1][i]; A[i * 7 + 4] = col[2][i];
            }
        } else if (g == 1) {
            for (int i = 0; i < NX; ++i) { A[i * 7 + 5] = col[0][i]; A[i * 7 + 6] = col[1][i]; }
        } else {
            for (int i = 0; i < NX; ++i) { Bm[i * 2] = col[0][i]; Bm[i * 2 + 1] = col[1][i]; }
        }
    }
}

// arg-min over cost[0..B): one block; ties -> lowest index; NaN treated as +inf
__global__ __launch_bounds__(256) void admpc_argmin_kernel(const double* __restrict__ cost, int B, int64_t offset,
                                                           double* __restrict__ val, int64_t* __restrict__ idx)
{
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    double best = INFINITY; int64_t bi = INT64_MAX;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        double c = cost[i]; if (!(c == c)) c = INFINITY;
        if (c < best || (c == best && (int64_t)i < bi)) { best = c; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(best, o, WAVE);
        int64_t oi = __shfl_xor((long long)bi, o, WAVE);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int w = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) if (sv[i] < best || (sv[i] == best && si[i] < bi)) { best = sv[i]; bi = si[i]; }
        *val = best;
        *idx = (bi == INT64_MAX ? 0 : bi) + offset;
    }
}

// post-solve epilogue: validity test ad_3d_optimizer.py:385-394 + Ackermann mapping
// create_ros_ad_mpc.py:95-98; one thread per instance
__global__ void admpc_epilogue_kernel(int N, int B, const double* __restrict__ xopt, const double* __restrict__ uopt,
                                      const double* __restrict__ xref_xy, float* __restrict__ ack, int32_t* __restrict__ valid)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* x = xopt + (size_t)b * (N + 1) * NX;
    const double* r = xref_xy + (size_t)b * (N + 1) * 2;
    const int n = N + 1;
    double s = 0.0, mx = 0.0;
    for (int i = 0; i < n - 1; ++i) {
        const double dxv = r[i * 2] - x[i * 7], dyv = r[i * 2 + 1] - x[i * 7 + 1];
        const double d = sqrt(dxv * dxv + dyv * dyv);
        s += d; mx = fmax(mx, d);
    }
    const double mean = s / n;
    double var = 0.0;
    for (int i = 0; i < n; ++i) {
        double d = 0.0;
        if (i < n - 1) { const double dxv = r[i * 2] - x[i * 7], dyv = r[i * 2 + 1] - x[i * 7 + 1]; d = sqrt(dxv * dxv + dyv * dyv); }
        var += (d - mean) * (d - mean);
    }
    var /= (n - 1);
    valid[b] = (mean < 3.0 && var < 2.0 && mx < 4.0) ? 1 : 0;
    const double* u = uopt + (size_t)b * N * NU;
    ack[b * 4 + 0] = (float)x[6]; ack[b * 4 + 1] = (float)u[1]; ack[b * 4 + 2] = (float)x[3]; ack[b * 4 + 3] = (float)u[0];
}

}  // namespace

// =============================================================================================
// C ABI (include/admpc.h)
// =============================================================================================
#include <string>
#include <cstdio>
#include <cstring>

struct AdmpcSolver {
    AdmpcConfig cfg;
    AdmpcConfig* d_cfg;
    int device;
    int num_cu;
    int lds_bytes;
    int blocks_per_cu;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(ADMPC_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" {

const char* admpc_last_error(void) { return g_err.c_str(); }
const char* admpc_version(void) { return "admpc-mi355x 0.1 (gfx950)"; }

int admpc_default_config(AdmpcConfig* c, int N, double Ts)
{
    if (!c || N < 2 || N > ADMPC_MAX_N || !(Ts > 0)) return fail(ADMPC_EINVAL, "admpc_default_config: bad N/Ts");
    memset(c, 0, sizeof *c);
    c->N = N; c->ipm_iter_max = 50; c->sqp_iters = 1; c->n_gp = 0; c->Ts = Ts;
    const double q[NX] = {10, 10, 100, 0, 0, 0, 0}, r[NU] = {1, 100};          // create_ros_ad_mpc.py:58-59
    for (int i = 0; i < NX; ++i) { c->W[i] = q[i]; c->We[i] = q[i] * 1e-6; }   // ad_3d_optimizer.py:149-151
    for (int j = 0; j < NU; ++j) c->W[NX + j] = r[j];
    c->lbu[0] = -10; c->lbu[1] = -3; c->ubu[0] = 5; c->ubu[1] = 3;             // ad_3d.py:66-71
    c->lbx_delta = -0.52; c->ubx_delta = 0.52;
    c->zl = c->zu = 10;                                                        // ad_3d_optimizer.py:171-173
    const double mass = 1500, f_mass = 900, r_mass = mass - f_mass, Lw = 2.7;  // ad_3d.py:47-60
    c->mass = mass; c->L_F = Lw * (1 - f_mass / mass); c->L_R = Lw * (1 - r_mass / mass);
    c->Iz = c->L_F * c->L_R * (r_mass + f_mass);
    c->Cf = f_mass * 0.5 * 9.81 * 0.165 * 180 / 3.14195; c->Cr = r_mass * 0.5 * 9.81 * 0.165 * 180 / 3.14195;
    c->ipm_mu0 = 1.0; c->ipm_thr0 = 0.1; c->ipm_tol_comp = 1e-10; c->ipm_tol_res = 1e-9; c->ipm_tol_step = 1e-6;
    return ADMPC_OK;
}

static int validate(const AdmpcConfig* c)
{
    if (c->N < 2 || c->N > ADMPC_MAX_N) return fail(ADMPC_EINVAL, "N out of range [2,128]");
    if (!(c->Ts > 0)) return fail(ADMPC_EINVAL, "Ts must be positive");
    if (c->n_gp < 0 || c->n_gp > ADMPC_GP_MAX) return fail(ADMPC_EINVAL, "n_gp out of range");
    for (int g = 0; g < c->n_gp; ++g) {
        const AdmpcGp& gp = c->gp[g];
        if (gp.out < 3 || gp.out > 5 || gp.feat < 3 || gp.feat > 8 || gp.n_points < 0 || gp.n_points > ADMPC_GP_MAX_POINTS)
            return fail(ADMPC_EINVAL, "GP: out must be in {3,4,5}, feat in {3..8}, n_points <= 32");
    }
    if (!(c->W[NX] > 0 && c->W[NX + 1] > 0)) return fail(ADMPC_EINVAL, "input weights must be positive (strict convexity)");
    if (c->ipm_iter_max < 1) return fail(ADMPC_EINVAL, "ipm_iter_max < 1");
    return ADMPC_OK;
}

int admpc_create(const AdmpcConfig* cfg, int device, AdmpcSolver** out)
{
    if (!cfg || !out) return fail(ADMPC_EINVAL, "admpc_create: null argument");
    int rc = validate(cfg); if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ADMPC_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return fail(ADMPC_ENODEV, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    AdmpcSolver* s = new (std::nothrow) AdmpcSolver();
    if (!s) return fail(ADMPC_ENOMEM, "out of host memory");
    s->cfg = *cfg; s->device = device; s->num_cu = prop.multiProcessorCount;
    s->lds_bytes = lds_doubles(cfg->N) * (int)sizeof(double);
    if (s->lds_bytes > 160 * 1024) { delete s; return fail(ADMPC_EINVAL, "horizon too long for the LDS-resident kernel"); }
    s->blocks_per_cu = (160 * 1024) / s->lds_bytes;
    if (s->blocks_per_cu > 16) s->blocks_per_cu = 16;
    if (s->blocks_per_cu < 1) s->blocks_per_cu = 1;
    hipError_t e = hipMalloc((void**)&s->d_cfg, sizeof(AdmpcConfig));
    if (e != hipSuccess) { delete s; return fail(ADMPC_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemcpy(s->d_cfg, cfg, sizeof(AdmpcConfig), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(s->d_cfg); delete s; return fail(ADMPC_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    // opt in to > 64 KB of dynamic LDS
    const void* kerns[3] = { (const void*)admpc_solve_kernel<1>, (const void*)admpc_solve_kernel<2>, (const void*)admpc_solve_kernel<4> };
    for (int i = 0; i < 3; ++i) (void)hipFuncSetAttribute(kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *out = s;
    return ADMPC_OK;
}

void admpc_destroy(AdmpcSolver* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipFree(s->d_cfg);
    delete s;
}

int admpc_solve_batch(AdmpcSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* p,
                      double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream)
{
    if (!s) return fail(ADMPC_EINVAL, "null solver");
    if (B < 0) return fail(ADMPC_EINVAL, "negative batch");
    if (B == 0) return ADMPC_OK;
    if (!x0 || !yref || !yref_e || !p || !xbar || !ubar) return fail(ADMPC_EINVAL, "null array argument");
    HIPCHK(hipSetDevice(s->device));
    const int N = s->cfg.N;
    int grid = s->num_cu * s->blocks_per_cu;
    if (grid > B) grid = B;
    hipStream_t st = (hipStream_t)stream;
    if (2 * N <= 64) hipLaunchKernelGGL(admpc_solve_kernel<1>, dim3(grid), dim3(WAVE), s->lds_bytes, st, s->d_cfg, B, x0, yref, yref_e, p, xbar, ubar, cost, status, iters);
    else if (2 * N <= 128) hipLaunchKernelGGL(admpc_solve_kernel<2>, dim3(grid), dim3(WAVE), s->lds_bytes, st, s->d_cfg, B, x0, yref, yref_e, p, xbar, ubar, cost, status, iters);
    else return fail(ADMPC_EINVAL, "N > 64 not supported by this build");
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_shoot_batch(AdmpcSolver* s, int B, const double* xbar, const double* ubar, const double* p,
                      double* phi, double* A, double* Bm, void* stream)
{
    if (!s || B < 0) return fail(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!xbar || !ubar || !p || !phi || !A || !Bm) return fail(ADMPC_EINVAL, "null array argument");
    HIPCHK(hipSetDevice(s->device));
    long total = (long)B * s->cfg.N * 3;
    int grid = (int)((total + WAVE - 1) / WAVE); if (grid > s->num_cu * 32) grid = s->num_cu * 32;
    hipLaunchKernelGGL(admpc_shoot_kernel, dim3(grid), dim3(WAVE), 0, (hipStream_t)stream, s->d_cfg, B, xbar, ubar, p, phi, A, Bm);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_argmin(AdmpcSolver* s, const double* cost, int B, int64_t index_offset, double* val, int64_t* idx, void* stream)
{
    if (!s || !cost || !val || !idx || B <= 0) return fail(ADMPC_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(s->device));
    hipLaunchKernelGGL(admpc_argmin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cost, B, index_offset, val, idx);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_epilogue_batch(AdmpcSolver* s, int B, const double* xopt, const double* uopt, const double* xref_xy,
                         float* ack, int32_t* valid, void* stream)
{
    if (!s || B < 0) return fail(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!xopt || !uopt || !xref_xy || !ack || !valid) return fail(ADMPC_EINVAL, "null array argument");
    HIPCHK(hipSetDevice(s->device));
    hipLaunchKernelGGL(admpc_epilogue_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->cfg.N, B, xopt, uopt, xref_xy, ack, valid);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

}  // extern "C"
