// model_dev.h -- device code of the vehicle model shared by the translation units of the library: f(x, u, p) with the non-zero
// pattern of its Jacobians (ad_3d_optimizer.py:280-310), the GP residual (gp.py:81-138), the bounded-argument sincos / exp, and one
// ERK4 step with forward sensitivities for a group of columns (acados ERK, acados_solver_sim_car.c:655-665).
// Include INSIDE the translation unit's anonymous namespace, after NX / NU / NY are defined; every function is __forceinline__.
#pragma once

// ---------------------------------------------------------------------------------------------
// model: f and the non-zero pattern of its Jacobians (ad_3d_optimizer.py:280-310)
// ---------------------------------------------------------------------------------------------
template <class T>
struct ModelEvalT {
    T f[NX];
    T j0[3], j1[3];           // rows 0,1 of Jx over (psi, vx, vy)
    T a[3][4];                // rows 3,4,5 of Jx over (vx, vy, psi_dot, delta)
    T bu[3][2];               // rows 3,4,5 of Ju
};

// sin and cos of a moderate argument (|x| < ~1e4: yaw angles, steering angles) in ~35 fp64 instructions: Cody-Waite reduction by
// pi/2 in two parts and the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4] (error < 1 ulp there).  ocml's
// sincos spends ~100 instructions, most of them on a reduction for huge arguments that cannot occur here; the model is
// evaluated 4 times per stage and thread, each with two sincos.  Non-finite input gives NaN (as libm).
__device__ __forceinline__ void sincos_small(const double x, double* sn, double* cs) {
    const double n = rint(x * 6.36619772367581382433e-01);                 // 2 / pi
    double y = fma(-n, 1.57079632673412561417e+00, x);                    // pi/2, first 33 bits
    y = fma(-n, 6.07710050650619224932e-11, y);                           // pi/2 - first part
    const double z = y * y;
    double rs = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    rs = fma(z, rs, 2.75573137070700676789e-06); rs = fma(z, rs, -1.98412698298579493134e-04); rs = fma(z, rs, 8.33333333332248946124e-03);
    const double s0 = fma(y * z, fma(z, rs, -1.66666666666666324348e-01), y);
    double rc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    rc = fma(z, rc, -2.75573143513906633035e-07); rc = fma(z, rc, 2.48015872894767294178e-05); rc = fma(z, rc, -1.38888888888741095749e-03);
    rc = fma(z, rc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c0 = w + (((1.0 - w) - hz) + z * z * rc);
    const int q = (int)n & 3;
    const double sq = (q & 1) ? c0 : s0, cq = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -sq : sq;
    *cs = ((q + 1) & 2) ? -cq : cq;
}
// fp32: the library routine (1-2 ulp over the whole range)
__device__ __forceinline__ void sincos_small(const float x, float* sn, float* cs) { sincosf(x, sn, cs); }

// exp(x) for x <= 0 (the squared-exponential kernel): x = k ln2 + r, |r| <= ln2 / 2, Taylor polynomial of degree 13 (remainder
// < 4e-18), v_ldexp_f64.  ~20 instructions; underflows to 0 like libm.
__device__ __forceinline__ double exp_nonpos(const double x) {
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                                      // 1 / 13!
    p = fma(p, r, 2.08767569878681e-09);  p = fma(p, r, 2.505210838544172e-08); p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06); p = fma(p, r, 2.48015873015873e-05);  p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.3888888888888889e-03); p = fma(p, r, 8.333333333333333e-03); p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p, (int)fmax(k, -1100.0));
}
__device__ __forceinline__ float exp_nonpos(const float x) { return expf(x); }

// Mean and gradient of one squared-exponential GP over 1..3 features (anisotropic length scale, gp.py:81-138).  The three
// threads of a stage (adjacent lanes 3m, 3m+1, 3m+2: the kernel maps 63 tasks to a wave) integrate the same state, so they evaluate the same GP: each takes every third training
// point and the partial sums are combined by lane shuffles, in the same order on all three lanes (identical results).
template <class T>
__device__ __forceinline__ void gp_eval(const AdmpcGp& g, const T (&z)[ADMPC_GP_MAX_FEAT], T& mu, T (&dmu)[ADMPC_GP_MAX_FEAT]) {
    T m = 0, d0 = 0, d1 = 0, d2 = 0;
    const int n = g.n_points, nf = g.n_feat;
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));     // lane id (the callers run one wave per block)
    asm volatile("" : "+v"(lane));            // recomputed at every call: hoisted out of a persistent loop these three values were parked in scratch
    const int sub = lane - 3 * (int)(((unsigned)lane * 21846u) >> 16), base = lane - sub;   // lane % 3 without a narrow urem (see div7)
    const T sf = (T)g.sigma_f;
    const T il0 = (T)g.inv_l2[0];
    if (nf == 1) {                          // wave-uniform: the one-feature regressors of config 3 keep their short loop
        for (int i = sub; i < n; i += 3) {
            const T e0 = z[0] - (T)g.Z[0][i];
            const T ka = sf * exp_nonpos((T)-0.5 * e0 * e0 * il0) * (T)g.alpha[i];
            m += ka;
            d0 -= ka * e0 * il0;
        }
    } else {
        const T il1 = (T)g.inv_l2[1], il2 = nf > 2 ? (T)g.inv_l2[2] : (T)0;   // unused feature: weight 0
        for (int i = sub; i < n; i += 3) {
            const T e0 = z[0] - (T)g.Z[0][i], e1 = z[1] - (T)g.Z[1][i], e2 = z[2] - (T)g.Z[2][i];
            const T ka = sf * exp_nonpos((T)-0.5 * (e0 * e0 * il0 + e1 * e1 * il1 + e2 * e2 * il2)) * (T)g.alpha[i];
            m += ka;
            d0 -= ka * e0 * il0; d1 -= ka * e1 * il1; d2 -= ka * e2 * il2;
        }
    }
    auto tri = [&](T v) { const T a = __shfl(v, base), b = __shfl(v, base + 1), c = __shfl(v, base + 2); return (a + b) + c; };
    mu = tri(m) + (T)g.ymean; dmu[0] = tri(d0); dmu[1] = (T)0; dmu[2] = (T)0;
    if (nf > 1) { dmu[1] = tri(d1); dmu[2] = tri(d2); }
}

// T = float: the reference's "+ 1e-99" in the slip-angle denominators (ad_3d_optimizer.py:290,296-297) is 0 in fp32, and with
// v_x = 0 the dynamic branch would be inf * 0 = NaN even when the blend parameter p switches it off.  The fp32 instantiation
// therefore drops the dynamic branch altogether when p == 0 (the shipped blend speeds: pure kinematic model) and keeps the
// blended value for 0 < p <= 1 (hard parts).
template <class T>
__device__ __forceinline__ void model_eval(const AdmpcConfig* __restrict__ c, const T* x, const T* u, T p, ModelEvalT<T>& e)
{
    const T psi = x[2], vx = x[3], vy = x[4], r = x[5], dl = x[6];
    const T m = (T)c->mass, LF = (T)c->L_F, LR = (T)c->L_R, Iz = (T)c->Iz, Cf = (T)c->Cf, Cr = (T)c->Cr;
    const T L = LR + LF;
    // The wheelbase enters by its reciprocals: every `/ L` of the reference's expressions (ad_3d_optimizer.py:299, 306) is one IEEE
    // division sequence of ~11 dependent fp64 instructions per lane, and there were thirteen of them per evaluation -- 64 % of the
    // vector instructions of the shooting phase.  iL and LRL depend on the configuration only (one division each per phase: the four
    // evaluations of a step share them); the products differ from the quotients in the last bit.
    const T iL = (T)1 / L, LRL = LR / L;
    T sp, cp, sd, cd;
    sincos_small(psi, &sp, &cp);
    sincos_small(dl, &sd, &cd);
    e.f[0] = vx * cp - vy * sp;
    e.f[1] = vx * sp + vy * cp;
    e.f[2] = r;
    e.j0[0] = -vx * sp - vy * cp; e.j0[1] = cp; e.j0[2] = -sp;
    e.j1[0] = vx * cp - vy * sp;  e.j1[1] = sp; e.j1[2] = cp;
    const T v = vx + (T)1e-99;
    T iv = (T)1 / v;
    if constexpr (sizeof(T) == 4) iv = p == (T)0 ? (T)0 : iv;
    const T Ffy = 2 * Cf * (dl - (vy + LF * r) * iv);
    const T Fry = 2 * Cr * (LR * r - vy) * iv;
    const T im = (T)1 / m, iIz = (T)1 / Iz;
    const T kk = u[1] * vx + dl * u[0];
    const T dyn3 = u[0] - im * Ffy * sd + vy * r;
    const T dyn4 = im * (Fry + Ffy * cd) - vx * r;
    const T dyn5 = iIz * (LF * Ffy * cd - LR * Fry);
    const T q = (T)1 - p;
    e.f[3] = p * dyn3 + q * u[0];
    e.f[4] = p * dyn4 + q * (kk * LRL);
    e.f[5] = p * dyn5 + q * (kk * iL);
    e.f[6] = u[1];
    const T gF[4] = { 2 * Cf * (vy + LF * r) * iv * iv, -2 * Cf * iv, -2 * Cf * LF * iv, 2 * Cf };
    const T gR[4] = { -Fry * iv, -2 * Cr * iv, 2 * Cr * LR * iv, (T)0 };
    T d3[4], d4[4], d5[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        d3[i] = -gF[i] * sd * im;
        d4[i] = (gR[i] + gF[i] * cd) * im;
        d5[i] = (LF * gF[i] * cd - LR * gR[i]) * iIz;
    }
    d3[1] += r;  d3[2] += vy;  d3[3] += -Ffy * cd * im;
    d4[0] += -r; d4[2] += -vx; d4[3] += -Ffy * sd * im;
    d5[3] += -LF * Ffy * sd * iIz;
    const T k4[4] = { u[1] * LRL, (T)0, (T)0, u[0] * LRL };
    const T k5[4] = { u[1] * iL, (T)0, (T)0, u[0] * iL };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e.a[0][i] = p * d3[i];
        e.a[1][i] = p * d4[i] + q * k4[i];
        e.a[2][i] = p * d5[i] + q * k5[i];
    }
    e.bu[0][0] = (T)1;              e.bu[0][1] = (T)0;
    e.bu[1][0] = q * dl * LRL;  e.bu[1][1] = q * vx * LRL;
    e.bu[2][0] = q * dl * iL;   e.bu[2][1] = q * vx * iL;
    const int ngp = c->n_gp;
    for (int g = 0; g < ngp; ++g) {          // residual GPs: out in {3,4,5}, feat in {3..8} (validated on the host)
        const AdmpcGp& gp = c->gp[g];
        const int out = gp.out - 3, nf = gp.n_feat;
        // static indexing only: runtime-indexed private arrays would live in scratch memory
        int fd[ADMPC_GP_MAX_FEAT];
        T z[ADMPC_GP_MAX_FEAT], dmu[ADMPC_GP_MAX_FEAT], mu;
#pragma unroll
        for (int d = 0; d < ADMPC_GP_MAX_FEAT; ++d) {
            fd[d] = d < nf ? gp.feat[d] - 3 : -1;                // 0..3 -> (vx,vy,r,delta), 4..5 -> (u0,u1); -1: unused
            z[d] = fd[d] == 0 ? vx : fd[d] == 1 ? vy : fd[d] == 2 ? r : fd[d] == 3 ? dl : fd[d] == 4 ? u[0] : fd[d] == 5 ? u[1] : (T)0;
        }
        gp_eval<T>(gp, z, mu, dmu);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const bool so = out == o;
            e.f[3 + o] += so ? mu : (T)0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) e.a[o][cc] += (so && fd[0] == cc) ? dmu[0] : (T)0;
            e.bu[o][0] += (so && fd[0] == 4) ? dmu[0] : (T)0;
            e.bu[o][1] += (so && fd[0] == 5) ? dmu[0] : (T)0;
        }
        if (nf > 1) {                        // wave-uniform: one-feature regressors skip the selects of the other two features
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const bool so = out == o;
#pragma unroll
                for (int d = 1; d < ADMPC_GP_MAX_FEAT; ++d) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) e.a[o][cc] += (so && fd[d] == cc) ? dmu[d] : (T)0;
                    e.bu[o][0] += (so && fd[d] == 4) ? dmu[d] : (T)0;
                    e.bu[o][1] += (so && fd[d] == 5) ? dmu[d] : (T)0;
                }
            }
        }
    }
}

// d(column)/dt = Jx * s (+ Ju column for an input column)
template <class T>
__device__ __forceinline__ void sens_rhs(const ModelEvalT<T>& e, const T* s, int ucol, T* d)
{
    d[0] = e.j0[0] * s[2] + e.j0[1] * s[3] + e.j0[2] * s[4];
    d[1] = e.j1[0] * s[2] + e.j1[1] * s[3] + e.j1[2] * s[4];
    d[2] = s[5];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T a = e.a[r][0] * s[3] + e.a[r][1] * s[4] + e.a[r][2] * s[5] + e.a[r][3] * s[6];
        a += ucol == 0 ? e.bu[r][0] : (ucol == 1 ? e.bu[r][1] : (T)0);
        d[3 + r] = a;
    }
    d[6] = ucol == 1 ? (T)1 : (T)0;
}

// One ERK4 step of length h for the state and for the NC sensitivity columns of group g:
//   g=0: x-columns 2,3,4   g=1: x-columns 5,6   g=2: u-columns 0,1
// Results: phi[7] (all groups), col[c][7] = column c of the group of A (g<2) or B (g=2).
template <class T>
__device__ __forceinline__ void rk4_group(const AdmpcConfig* __restrict__ c, const T* x, const T* u, T p, T h,
                                          int g, T* phi, T col[3][NX])
{
    const int xcol0 = g == 0 ? 2 : 5;
    T kx[NX], accx[NX];
    T kS[3][NX], accS[3][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { kx[i] = (T)0; accx[i] = (T)0; }
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int i = 0; i < NX; ++i) { kS[cc][i] = (T)0; accS[cc][i] = (T)0; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const T cs = (s == 0) ? (T)0 : (s == 3 ? (T)1 : (T)0.5);
        const T ws = (s == 0 || s == 3) ? (T)(1.0 / 6.0) : (T)(2.0 / 6.0);
        T X[NX], S[3][NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) X[i] = x[i] + cs * h * kx[i];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T id = (g < 2 && i == xcol0 + cc) ? (T)1 : (T)0;
                S[cc][i] = id + cs * h * kS[cc][i];
            }
        ModelEvalT<T> e;
        model_eval<T>(c, X, u, p, e);
#pragma unroll
        for (int i = 0; i < NX; ++i) { kx[i] = e.f[i]; accx[i] += ws * e.f[i]; }
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            sens_rhs<T>(e, S[cc], g == 2 ? cc : -1, kS[cc]);
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
            T id = (g < 2 && i == xcol0 + cc) ? (T)1 : (T)0;
            col[cc][i] = id + h * accS[cc][i];
        }
}

