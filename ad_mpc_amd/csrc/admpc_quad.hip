// admpc_quad.hip -- the second vehicle model behind the engine: one SQP-RTI step of the reference's quadrotor MPC
// (src/quad_mpc/quad_3d_optimizer.py:150-207 formulation, :341-393 dynamics; nx = 13, nu = 4, N = 10) for a batch of instances.
//
// One workgroup of one wavefront per instance (grid-stride over instances); the whole step lives in LDS and registers:
//   1. shooting: 17 N tasks (stage, sensitivity column) spread over the 64 lanes -- each integrates the state and ONE column of
//      [S_x S_u] through classic RK4 with the model's forward-mode tangent (the model is polynomial: no Jacobian matrix is formed);
//   2. condensing: lane i <-> input i = (stage, rotor) carries column i of Gamma_k = d x_k / d u through the stages; the tracking
//      terms Gamma' Q Gamma accumulate into the dense Hessian in LDS (the reference solves the same condensed QP: FULL_CONDENSING_HPIPM);
//   3. box-constrained dense QP (N nu <= 64 inputs, one per lane): Mehrotra predictor-corrector.  N nu = 40 (the shipped N = 10): the
//      Newton systems go through dense40.h, the machinery of the car's condensed kernel (register-resident LDL' with DPP rank-1
//      updates, generated substitution assembly, DPP symmetric mat-vec); the Hessian row of a lane is accumulated in registers with
//      the other lanes' Gamma picked up by DPP broadcasts.  Other horizons: Cholesky in LDS with row i on lane i (generic path);
//   4. expansion of the states, full step, cost, status.
// The generic path follows oracle/quad_oracle.c operation by operation where the order matters (Cholesky, forward substitution); the
// fast path reaches the same iterates (identical iteration counts on every tested batch).  DESIGN section 7 has the numbers.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <new>
#include <type_traits>
#include "../../include/admpc.h"
#include "../../include/admpc_quad.h"

extern "C" int admpc_set_error(int code, const char* msg);          // admpc_kernels.hip: thread-local message behind admpc_last_error()

#ifdef ADMPC_QUAD_TIMERS
// per-wave accumulators, flushed once at the end of the kernel (an atomic per stamp costs more than most phases)
__device__ unsigned long long g_quad_ticks[16];
#define QDECL() unsigned long long qacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, qlast = 0
#define QSTAMP(id) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); qacc[id] += t_ - qlast; qlast = t_; } while (0)
#define QSTART() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(qlast))
#define QFLUSH() do { if (threadIdx.x == 0) for (int q_ = 0; q_ < 16; ++q_) atomicAdd(&g_quad_ticks[q_], qacc[q_]); } while (0)
#else
#define QDECL() do { } while (0)
#define QSTAMP(id) do { } while (0)
#define QSTART() do { } while (0)
#define QFLUSH() do { } while (0)
#endif

namespace {

constexpr int QX = ADMPC_QUAD_NX, QU = ADMPC_QUAD_NU, QY = ADMPC_QUAD_NY;
typedef AdmpcQuadConfig Cfg;

// f(x, u) and df = Jx sx + Ju su   (oracle/quad_oracle.c:quad_f_tan, same expressions)
// rotation matrix of a quaternion and its directional derivative along sq (utils.py:323-338)
__device__ __forceinline__ void quad_rot(double qw, double qx, double qy, double qz, double (&R)[3][3]) {
    R[0][0] = 1 - 2 * (qy * qy + qz * qz); R[0][1] = 2 * (qx * qy - qw * qz); R[0][2] = 2 * (qx * qz + qw * qy);
    R[1][0] = 2 * (qx * qy + qw * qz); R[1][1] = 1 - 2 * (qx * qx + qz * qz); R[1][2] = 2 * (qy * qz - qw * qx);
    R[2][0] = 2 * (qx * qz - qw * qy); R[2][1] = 2 * (qy * qz + qw * qx); R[2][2] = 1 - 2 * (qx * qx + qy * qy);
}
__device__ __forceinline__ void quad_drot(double qw, double qx, double qy, double qz, double sw, double sxx, double sy, double sz, double (&D)[3][3]) {
    D[0][0] = -4 * (qy * sy + qz * sz); D[0][1] = 2 * (sxx * qy + qx * sy - sw * qz - qw * sz); D[0][2] = 2 * (sxx * qz + qx * sz + sw * qy + qw * sy);
    D[1][0] = 2 * (sxx * qy + qx * sy + sw * qz + qw * sz); D[1][1] = -4 * (qx * sxx + qz * sz); D[1][2] = 2 * (sy * qz + qy * sz - sw * qx - qw * sxx);
    D[2][0] = 2 * (sxx * qz + qx * sz - sw * qy - qw * sy); D[2][1] = 2 * (sy * qz + qy * sz + sw * qx + qw * sxx); D[2][2] = -4 * (qx * sxx + qy * sy);
}

// trig / gq: the first node's GP-state parameter (quad_3d_optimizer.py:291-297, :546-552; oracle: gpx).  On lanes with trig set the GP
// features and the rotation of the means come from the constant state gq[13] (zero tangent along sx), elsewhere from x.  gq must be
// readable on every lane when c->n_gp > 0.
__device__ __forceinline__ void quad_f_tan(const Cfg* __restrict__ c, const double* x, const double* u, const double* sx, const double* su,
                                           bool trig, const double* __restrict__ gq, double* f, double* df)
{
    const double qw = x[3], qx = x[4], qy = x[5], qz = x[6], r0 = x[10], r1 = x[11], r2 = x[12];
    const double sw = sx[3], sxx = sx[4], sy = sx[5], sz = sx[6], t0 = sx[10], t1 = sx[11], t2 = sx[12];
    double ga[3] = { 0, 0, 0 }, dga[3] = { 0, 0, 0 };        // GP residual and drag term of the acceleration, evaluated first (shorter live ranges)
    if (c->n_gp > 0) {                       // wave-uniform.  GP residual: v' += R(q) mu(z), z = [x with v in the body frame; u]
        // the state the features and the rotation come from: entries 3..12 (attitude, velocity, body rates) of x or of the parameter
        double xe[10], se[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            double gv = gq[3 + i];
            asm volatile("" : "+v"(gv));      // a value, not an address: otherwise the select becomes `load (trig ? gq : &x)` with x spilled to scratch for it
            xe[i] = trig ? gv : x[3 + i]; se[i] = trig ? 0.0 : sx[3 + i];
        }
        const double ew = xe[0], ex = xe[1], ey = xe[2], ez = xe[3], fw = se[0], fx = se[1], fy = se[2], fz = se[3];
        const double R[3][3] = { { 1 - 2 * (ey * ey + ez * ez), 2 * (ex * ey - ew * ez), 2 * (ex * ez + ew * ey) },
                                 { 2 * (ex * ey + ew * ez), 1 - 2 * (ex * ex + ez * ez), 2 * (ey * ez - ew * ex) },
                                 { 2 * (ex * ez - ew * ey), 2 * (ey * ez + ew * ex), 1 - 2 * (ex * ex + ey * ey) } };
        // dR = directional derivative of R, formed where it is used (twice) instead of kept live across the GP loop
        auto dRm = [&](double (&D)[3][3]) {
            D[0][0] = -4 * (ey * fy + ez * fz); D[0][1] = 2 * (fx * ey + ex * fy - fw * ez - ew * fz); D[0][2] = 2 * (fx * ez + ex * fz + fw * ey + ew * fy);
            D[1][0] = 2 * (fx * ey + ex * fy + fw * ez + ew * fz); D[1][1] = -4 * (ex * fx + ez * fz); D[1][2] = 2 * (fy * ez + ey * fz - fw * ex - ew * fx);
            D[2][0] = 2 * (fx * ez + ex * fz - fw * ey - ew * fy); D[2][1] = 2 * (fy * ez + ey * fz + fw * ex + ew * fx); D[2][2] = -4 * (ex * fx + ey * fy);
        };
        // candidate features: entries 7..16 of z (body-frame velocity, body rates, inputs); position and attitude are not offered
        double z[10], dz[10];
        { double dR[3][3]; dRm(dR);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a2 = 0, d2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { a2 += R[k][i] * xe[4 + k]; d2 += dR[k][i] * xe[4 + k] + R[k][i] * se[4 + k]; }
            z[i] = a2; dz[i] = d2;
            z[3 + i] = xe[7 + i]; dz[3 + i] = se[7 + i];
        } }
#pragma unroll
        for (int m = 0; m < QU; ++m) { z[6 + m] = u[m]; dz[6 + m] = su[m]; }
        double mb[3] = { 0, 0, 0 }, dmb[3] = { 0, 0, 0 };
        for (int g = 0; g < c->n_gp; ++g) {
            const AdmpcGp& gp = c->gp[g];
            // features by compare-select chains (a runtime-indexed private array would live in scratch memory)
            double zf[ADMPC_GP_MAX_FEAT], dzf[ADMPC_GP_MAX_FEAT], il[ADMPC_GP_MAX_FEAT];
#pragma unroll
            for (int k = 0; k < ADMPC_GP_MAX_FEAT; ++k) {
                const int fk = k < gp.n_feat ? gp.feat[k] - 7 : -1;
                double zv = 0, dzv = 0;
#pragma unroll
                for (int i = 0; i < 10; ++i) { zv = fk == i ? z[i] : zv; dzv = fk == i ? dz[i] : dzv; }
                zf[k] = zv; dzf[k] = dzv; il[k] = k < gp.n_feat ? gp.inv_l2[k] : 0.0;
            }
            double m = 0, dm = 0;
            for (int i = 0; i < gp.n_points; ++i) {
                double e = 0, de = 0;
#pragma unroll
                for (int k = 0; k < ADMPC_GP_MAX_FEAT; ++k) { const double dzk = zf[k] - gp.Z[k][i]; e += dzk * dzk * il[k]; de += dzk * il[k] * dzf[k]; }
                const double ka = gp.sigma_f * exp(-0.5 * e) * gp.alpha[i];
                m += ka; dm -= ka * de;
            }
            const int o = gp.out - 7;
#pragma unroll
            for (int i = 0; i < 3; ++i) { mb[i] += o == i ? m + gp.ymean : 0.0; dmb[i] += o == i ? dm : 0.0; }
        }
        double dR[3][3]; dRm(dR);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a2 = 0, d2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { a2 += R[i][k] * mb[k]; d2 += dR[i][k] * mb[k] + R[i][k] * dmb[k]; }
            ga[i] = a2; dga[i] = d2;
        }
    }
    if (c->rdrv[0] != 0.0 || c->rdrv[1] != 0.0 || c->rdrv[2] != 0.0) {      // wave-uniform.  Linear rotor drag (:364-381): v' += R(q) D R(q)' v
        const double R[3][3] = { { 1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy) },
                                 { 2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx) },
                                 { 2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy) } };
        const double dR[3][3] = { { -4 * (qy * sy + qz * sz), 2 * (sxx * qy + qx * sy - sw * qz - qw * sz), 2 * (sxx * qz + qx * sz + sw * qy + qw * sy) },
                                  { 2 * (sxx * qy + qx * sy + sw * qz + qw * sz), -4 * (qx * sxx + qz * sz), 2 * (sy * qz + qy * sz - sw * qx - qw * sxx) },
                                  { 2 * (sxx * qz + qx * sz - sw * qy - qw * sy), 2 * (sy * qz + qy * sz + sw * qx + qw * sxx), -4 * (qx * sxx + qy * sy) } };
        double wb[3], dwb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a2 = 0, d2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { a2 += R[k][i] * x[7 + k]; d2 += dR[k][i] * x[7 + k] + R[k][i] * sx[7 + k]; }
            wb[i] = c->rdrv[i] * a2; dwb[i] = c->rdrv[i] * d2;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a2 = 0, d2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { a2 += R[i][k] * wb[k]; d2 += dR[i][k] * wb[k] + R[i][k] * dwb[k]; }
            ga[i] += a2; dga[i] += d2;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { f[i] = x[7 + i]; df[i] = sx[7 + i]; }
    f[3] = 0.5 * (-r0 * qx - r1 * qy - r2 * qz);
    f[4] = 0.5 * ( r0 * qw + r2 * qy - r1 * qz);
    f[5] = 0.5 * ( r1 * qw - r2 * qx + r0 * qz);
    f[6] = 0.5 * ( r2 * qw + r1 * qx - r0 * qy);
    df[3] = 0.5 * (-t0 * qx - r0 * sxx - t1 * qy - r1 * sy - t2 * qz - r2 * sz);
    df[4] = 0.5 * ( t0 * qw + r0 * sw + t2 * qy + r2 * sy - t1 * qz - r1 * sz);
    df[5] = 0.5 * ( t1 * qw + r1 * sw - t2 * qx - r2 * sxx + t0 * qz + r0 * sz);
    df[6] = 0.5 * ( t2 * qw + r2 * sw + t1 * qx + r1 * sxx - t0 * qy - r0 * sy);
    const double a = c->max_thrust * (u[0] + u[1] + u[2] + u[3]) / c->mass;
    const double da = c->max_thrust * (su[0] + su[1] + su[2] + su[3]) / c->mass;
    const double c0 = 2 * (qx * qz + qw * qy), c1 = 2 * (qy * qz - qw * qx), c2 = 1 - 2 * (qx * qx + qy * qy);
    const double d0 = 2 * (sxx * qz + qx * sz + sw * qy + qw * sy), d1 = 2 * (sy * qz + qy * sz - sw * qx - qw * sxx),
                 d2 = -4 * (qx * sxx + qy * sy);
    f[7] = c0 * a; f[8] = c1 * a; f[9] = c2 * a - c->g;
    df[7] = d0 * a + c0 * da; df[8] = d1 * a + c1 * da; df[9] = d2 * a + c2 * da;
    double tx = 0, ty = 0, tz = 0, dtx = 0, dty = 0, dtz = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tx += c->max_thrust * u[i] * c->y_f[i]; ty -= c->max_thrust * u[i] * c->x_f[i]; tz += c->max_thrust * u[i] * c->z_l_tau[i];
        dtx += c->max_thrust * su[i] * c->y_f[i]; dty -= c->max_thrust * su[i] * c->x_f[i]; dtz += c->max_thrust * su[i] * c->z_l_tau[i];
    }
    f[10] = (tx + (c->J[1] - c->J[2]) * r1 * r2) / c->J[0];
    f[11] = (ty + (c->J[2] - c->J[0]) * r2 * r0) / c->J[1];
    f[12] = (tz + (c->J[0] - c->J[1]) * r0 * r1) / c->J[2];
    df[10] = (dtx + (c->J[1] - c->J[2]) * (t1 * r2 + r1 * t2)) / c->J[0];
    df[11] = (dty + (c->J[2] - c->J[0]) * (t2 * r0 + r2 * t0)) / c->J[1];
    df[12] = (dtz + (c->J[0] - c->J[1]) * (t0 * r1 + r0 * t1)) / c->J[2];
    f[7] += ga[0]; f[8] += ga[1]; f[9] += ga[2]; df[7] += dga[0]; df[8] += dga[1]; df[9] += dga[2];
}

// classic RK4, one step of length h: the state and ONE sensitivity column (col < 13: d/dx_col, else d/du_(col-13))
__device__ __forceinline__ void rk4_col(const Cfg* __restrict__ c, const double* x, const double* u, bool trig, const double* __restrict__ gq, double h, int col, double* phi, double* scol)
{
    const double cs[4] = { 0, 0.5, 0.5, 1.0 }, ws[4] = { 1.0 / 6, 2.0 / 6, 2.0 / 6, 1.0 / 6 };
    double kx[QX], ks[QX], ax[QX], as[QX], su[QU];
#pragma unroll
    for (int i = 0; i < QX; ++i) { kx[i] = 0; ks[i] = 0; ax[i] = 0; as[i] = 0; }
#pragma unroll
    for (int m = 0; m < QU; ++m) su[m] = col == QX + m ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        double X[QX], S[QX], f[QX], df[QX];
#pragma unroll
        for (int i = 0; i < QX; ++i) { X[i] = x[i] + cs[s] * h * kx[i]; S[i] = (col == i ? 1.0 : 0.0) + cs[s] * h * ks[i]; }
        quad_f_tan(c, X, u, S, su, trig, gq, f, df);
#pragma unroll
        for (int i = 0; i < QX; ++i) { kx[i] = f[i]; ks[i] = df[i]; ax[i] += ws[s] * f[i]; as[i] += ws[s] * df[i]; }
    }
#pragma unroll
    for (int i = 0; i < QX; ++i) { phi[i] = x[i] + h * ax[i]; scol[i] = (col == i ? 1.0 : 0.0) + h * as[i]; }
}

#include "dense40.h"      // the 40 x 40 LDL' / substitution machinery of the car's condensed kernel (fast path for N nu = 40)

// value of lane l (wave-uniform l): two v_readlane, no LDS round trip
__device__ __forceinline__ double bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// wave reductions: DPP scans (dense40.h); every lane receives the result
struct OpMin { static constexpr bool zf = false; static __device__ __forceinline__ double id() { return INFINITY; } static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); } };
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<OpSum>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<OpMaxNan>(v); }         // NaN-propagating
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<OpMin>(v); }

// LDS layout (doubles): A [N][13][13] | B [N][13][4] | b [N][13] | H, M: packed lower triangles n(n+1)/2 | gam [13][n] | vec [64] | xnew [(N+1)*13]
//   | xbs [(N+1)*13] the linearisation point | yrs [N*17 + 13] the references (stage, terminal): read stage by stage in the condensing and
//   expansion loops, where a global load would put a memory round trip into every stage | xhs [13] free response of the current stage
//   (the same for every lane: computed once, by lanes 0..12) | wts [26] stage and terminal weights (registers are the scarce resource)
struct Lds {
    double *A, *B, *b, *H, *M, *gam, *vec, *xnew, *xbs, *yrs, *xhs, *wts;
    __device__ Lds(double* p, int N) {
        const int n = N * QU, tri = n * (n + 1) / 2;
        A = p; B = A + N * QX * QX; b = B + N * QX * QU; H = b + N * QX; M = H + tri; gam = M + tri; vec = gam + QX * n; xnew = vec + 64;
        xbs = xnew + (N + 1) * QX; yrs = xbs + (N + 1) * QX; xhs = yrs + N * QY + QX; wts = xhs + QX;
    }
};
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }                    // j <= i
__device__ __forceinline__ int sym(int i, int j) { return i >= j ? tri(i, j) : tri(j, i); }
__host__ __device__ inline int quad_lds_doubles(int N) { const int n = N * QU; return N * (QX * QX + QX * QU + QX) + n * (n + 1) + QX * n + 64 + 2 * (N + 1) * QX + N * QY + QX + QX + 2 * QX; }

// gq: GP state of the first node (instance's x0 or the caller's gp_state); only read when the model carries GPs
__device__ void shoot_instance(const Cfg* __restrict__ c, const double* xb, const double* ub, const double* __restrict__ gq, const Lds& L, int lane, double* phi_out, int nt = 64)
{
    const int N = c->N;
    for (int t = lane; t < N * (QX + QU); t += nt) {
        const int k = t / (QX + QU), col = t - k * (QX + QU);
        double x[QX], u[QU], phi[QX], sc[QX];
#pragma unroll
        for (int i = 0; i < QX; ++i) x[i] = xb[k * QX + i];
#pragma unroll
        for (int m = 0; m < QU; ++m) u[m] = ub[k * QU + m];
        rk4_col(c, x, u, k == 0, gq, c->Ts, col, phi, sc);
#pragma unroll
        for (int i = 0; i < QX; ++i) {
            if (col < QX) L.A[(k * QX + i) * QX + col] = sc[i]; else L.B[(k * QX + i) * QU + (col - QX)] = sc[i];
            if (col == 0) { L.b[k * QX + i] = phi[i] - xb[(k + 1) * QX + i]; if (phi_out) phi_out[k * QX + i] = phi[i]; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void admpc_quad_shoot_kernel(const Cfg* __restrict__ c, int B, const double* __restrict__ xbarg, const double* __restrict__ ubarg,
                                                             const double* __restrict__ gpsg, double* __restrict__ phig, double* __restrict__ Ag, double* __restrict__ Bg)
{
    extern __shared__ double lds_raw[];
    const int N = c->N, lane = threadIdx.x;
    Lds L(lds_raw, N);
    for (int inst = blockIdx.x; inst < B; inst += gridDim.x) {
        shoot_instance(c, xbarg + (size_t)inst * (N + 1) * QX, ubarg + (size_t)inst * N * QU,
                       gpsg ? gpsg + (size_t)inst * QX : xbarg + (size_t)inst * (N + 1) * QX, L, lane, phig + (size_t)inst * N * QX);
        for (int i = lane; i < N * QX * QX; i += 64) Ag[(size_t)inst * N * QX * QX + i] = L.A[i];
        for (int i = lane; i < N * QX * QU; i += 64) Bg[(size_t)inst * N * QX * QU + i] = L.B[i];
        __syncthreads();
    }
}

// FAST (N nu = 40, the shipped horizon): the Newton systems go through dense40.h -- register-resident LDL' with DPP rank-1 updates and
// the generated substitution assembly instead of the LDS-resident Cholesky below (3.9x less time per interior-point iteration).
template <bool FAST>
__global__ __launch_bounds__(64) void admpc_quad_solve_kernel(const Cfg* __restrict__ c, int B, const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                             const double* __restrict__ yrefeg, double* __restrict__ xbarg, double* __restrict__ ubarg,
                                                             double* __restrict__ costg, int32_t* __restrict__ statusg, int32_t* __restrict__ itersg,
                                                             int* __restrict__ ticket, const double* __restrict__ gpsg, const int32_t* __restrict__ routeg, int which)
{
    extern __shared__ double lds_raw[];
    const int N = c->N, n = N * QU, lane = threadIdx.x;
    Lds L(lds_raw, N);
    const bool act = lane < n;
    const int li = act ? lane : 0, ji = li / QU, mi = li - ji * QU;
    const double Ts = c->Ts;
    // problem constants in registers / LDS: a read of *c inside a stage loop is a scalar load from global memory
    if (lane < QX) { L.wts[lane] = Ts * c->W[lane]; L.wts[QX + lane] = c->We[lane]; }
    const double Rw = Ts * c->W[QX + mi], lbm = c->lbu[mi], ubm = c->ubu[mi];
    const double thr0 = c->ipm_thr0, mu0 = c->ipm_mu0, tolc = c->ipm_tol_comp, tolr = c->ipm_tol_res;
    const int itmax = c->ipm_iter_max;
    const Dense40Lds W{L.H, L.M, L.gam, L.gam + 64};                // FAST: factor in the M region, exchange buffer / pivots in gam (free after condensing)
    if (FAST) { if (act) L.M[tri(li, li)] = 0.0; __syncthreads(); }     // diagonal slots of the packed unit factor: 0.0, never overwritten
    QDECL();
    // instances need 5 .. 18 interior-point iterations: with many rounds per wave the instances are drawn from a work counter (zeroed by
    // the host before the launch; + 5 % at B = 16384), with few a static stride is cheaper (no memset node, no atomics; + 5 % at B = 4096)
    // next instance of this wave: from the work counter (many rounds) or by a static stride
    auto next_inst = [&](int inst) -> int {
        if (!ticket) return inst + (int)gridDim.x;
        int tk = 0;
        if (lane == 0) tk = atomicAdd(ticket, 1);
        return (int)gridDim.x + __builtin_amdgcn_readfirstlane(tk);
    };
    for (int inst = blockIdx.x; inst < B;) {
        // routed solve (admpc_quad_solve_batch_routed): this handle carries the model of cluster `which`; the others' instances are left alone
        if (routeg && routeg[inst] != which) { inst = next_inst(inst); continue; }
        double* xb = xbarg + (size_t)inst * (N + 1) * QX;
        double* ub = ubarg + (size_t)inst * N * QU;
        const double* yr = yrefg + (size_t)inst * N * QY;
        const double* ye = yrefeg + (size_t)inst * QX;
        const double* x0 = x0g + (size_t)inst * QX;
        const double* gq = gpsg ? gpsg + (size_t)inst * QX : x0;       // GP state of the first node: run_optimization's default is the initial state
        QSTART();
        for (int i = lane; i < (N + 1) * QX; i += 64) L.xbs[i] = xb[i];
        for (int i = lane; i < N * QY + QX; i += 64) L.yrs[i] = i < N * QY ? yr[i] : ye[i - N * QY];
        // ---- 1. shooting
        shoot_instance(c, xb, ub, gq, L, lane, nullptr);
        QSTAMP(0);
        // ---- 2. condensing (oracle: condense)
        if constexpr (!FAST) { for (int j = 0; j <= li; ++j) if (act) L.H[tri(li, j)] = 0.0; }
        double hrow[FAST ? 40 : 1];                                  // FAST: row li of H in registers, Gamma of the other inputs through DPP broadcasts
#pragma unroll
        for (int j = 0; j < (FAST ? 40 : 1); ++j) hrow[j] = 0.0;
        double g[QX];
#pragma unroll
        for (int i = 0; i < QX; ++i) g[i] = 0.0;
        if (lane < QX) L.xhs[lane] = x0[lane] - xb[lane];
        const double ubar_i = ub[li];
        double grad = Rw * (ubar_i - yr[ji * QY + QX + mi]);
        __syncthreads();
        for (int k = 0; k < N; ++k) {
            double gn[QX];
            const int rr = lane < QX ? lane : 0;
            double xn = L.b[k * QX + rr];                                    // free response: row `lane` of A xh + b on lanes 0..12
#pragma unroll
            for (int cc = 0; cc < QX; ++cc) xn += L.A[(k * QX + rr) * QX + cc] * L.xhs[cc];
            // A_k is the same for every lane: its 169 entries are fetched by three coalesced loads (entry e on lane e mod 64) and handed
            // to the FMAs as scalar operands (v_readlane), instead of 169 broadcast reads of LDS per lane
            double Ar[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) { const int e = lane + 64 * m; Ar[m] = L.A[k * QX * QX + (e < QX * QX ? e : QX * QX - 1)]; }
#pragma unroll
            for (int r = 0; r < QX; ++r) {
                double s = 0.0;
#pragma unroll
                for (int cc = 0; cc < QX; ++cc) s += bcast(Ar[(r * QX + cc) / 64], (r * QX + cc) % 64) * g[cc];
                gn[r] = ji == k ? L.B[(k * QX + r) * QU + mi] : (ji < k ? s : 0.0);
            }
            QSTAMP(8);
#pragma unroll
            for (int r = 0; r < QX; ++r) { g[r] = gn[r]; if (act) L.gam[r * n + lane] = gn[r]; }
            if (lane < QX) L.xhs[lane] = xn;                                 // (LDS operations of one wave execute in order: every read above is done)
            __syncthreads();
            QSTAMP(9);
            const double* ref = L.yrs + (k + 1) * QY;                        // stage k + 1 (yrs[N * 17 ...] = terminal reference)
            const int lim = (k + 1) * QU;                                   // inputs of stages <= k
            double wg[QX];
#pragma unroll
            for (int cc = 0; cc < QX; ++cc) {
                const double wq = L.wts[(k + 1 < N ? 0 : QX) + cc];
                wg[cc] = g[cc] * wq;
                if (wq != 0.0) grad += wg[cc] * (L.xbs[(k + 1) * QX + cc] + L.xhs[cc] - ref[cc]);
            }
            if constexpr (FAST) {
                // H[li][j] += sum_cc wg[cc] Gamma_j[cc] for every j at once: Gamma_j[cc] = lane j's value, picked up inside the FMAs
                // (columns of inputs of later stages are zero on both sides)
                double Rb[QX][3];                                           // all 39 loads first: one LDS round trip per stage, not one per component
#pragma unroll
                for (int cc = 0; cc < QX; ++cc)
#pragma unroll
                    for (int m = 0; m < 3; ++m) Rb[cc][m] = L.gam[cc * n + 16 * m + (lane & 15)];
                static_for<0, 10>([&](auto qq) __attribute__((always_inline)) {
                    constexpr int i2 = 4 * decltype(qq)::value;
                    if (i2 < lim) {                                         // wave-uniform: the columns of later stages' inputs are still zero
                        static_for<0, QX>([&](auto ccc) __attribute__((always_inline)) {
                            constexpr int cc = decltype(ccc)::value;
                            fmac_rowbc4_ld<i2 % 16>(hrow[i2], hrow[i2 + 1], hrow[i2 + 2], hrow[i2 + 3], Rb[cc][i2 / 16], wg[cc]);
                        });
                    }
                });
            } else
            if (act && li < lim) {
                for (int j = 0; j <= li; ++j) {
                    double s = L.H[tri(li, j)];
#pragma unroll
                    for (int cc = 0; cc < QX; ++cc) s += wg[cc] * L.gam[cc * n + j];
                    L.H[tri(li, j)] = s;
                }
            }
            __syncthreads();
            QSTAMP(10);
        }
        if constexpr (FAST) {
            store_row_40(hrow, lds_byte_addr(L.H + (act ? tri(li, 0) : 0)));
            __syncthreads();
        }
        if (act) L.H[tri(li, li)] += Rw;
        __syncthreads();
        QSTAMP(1);
        // ---- 3. box QP (oracle: box_qp)
        const double lo = lbm - ubar_i, hi = ubm - ubar_i;
        double du = 0.0;
        double tl = act ? fmax(du - lo, thr0) : 1.0, tu = act ? fmax(hi - du, thr0) : 1.0;
        double ll = act ? mu0 / tl : 0.0, lu = act ? mu0 / tu : 0.0;
        double alpha_prev = 1.0;
        int it = 0, st = 0;
        bool cons = false;                                           // fallback mode (admpc_quad.h): no second-order term
        for (;; ++it) {
            int lz = lane;                                           // laundered lane id (dense40.h)
            asm volatile("" : "+v"(lz));
            double rs = grad - ll + lu;
            if constexpr (FAST) {
                const double hdu = dense40_symv(W, lane, act ? du : 0.0, lz);
                rs = act ? rs + hdu : 0.0;
            } else {
                L.vec[lane] = du;
                __syncthreads();
                if (act) for (int j = 0; j < n; ++j) rs += L.H[sym(li, j)] * L.vec[j];
            }
            const double rl = du - lo - tl, ru = hi - du - tu;
            const double mu = wave_sum(act ? tl * ll + tu * lu : 0.0) / (2.0 * n);
            const double cmax = wave_max(act ? fmax(tl * ll, tu * lu) : 0.0);
            const double rmax = wave_max(act ? fmax(fabs(rs), fmax(fabs(rl), fabs(ru))) : 0.0);
            __syncthreads();
            if (!(mu == mu) || !(rmax == rmax)) { st = 4; break; }
            if ((cmax <= tolc && rmax <= tolr) || it >= itmax + (cons ? ADMPC_QUAD_IPM_FALLBACK_ITER : 0)) break;
            if (!cons && it >= ADMPC_QUAD_IPM_FALLBACK_ITER) {          // still iterating (a limit cycle): start over, finish with plain centring steps
                cons = true;
                du = 0.0;
                tl = act ? fmax(du - lo, thr0) : 1.0; tu = act ? fmax(hi - du, thr0) : 1.0;
                ll = act ? mu0 / tl : 0.0; lu = act ? mu0 / tu : 0.0;
                alpha_prev = 1.0;
                --it;
                continue;                                              // this pass is redone from the cold start (the count is unchanged)
            }
            const double Dl = act ? ll / tl : 0.0, Du = act ? lu / tu : 0.0;
            if constexpr (!FAST) {
                if (act) { for (int j = 0; j <= li; ++j) L.M[tri(li, j)] = L.H[tri(li, j)]; L.M[tri(li, li)] += Dl + Du; }
                __syncthreads();
            }
            QSTAMP(2);
            // Cholesky M = L L' (lower), row i on lane i, columns left to right (oracle: chol)
            bool posdef = true;
            double myinv = 0.0;                                      // lane j: 1 / L_jj
            if constexpr (FAST) dense40_factorise(W, lane, act ? Dl + Du : 1.0, 0.0, lz);
            else
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                if (act && li >= j) {
                    const double* ri = L.M + tri(li, 0);
                    const double* rj = L.M + tri(j, 0);
                    s = ri[j];
                    int k2 = 0;
                    for (; k2 + 8 <= j; k2 += 8) {                   // eight products per round trip, subtracted in the oracle's order
                        double a[8], b[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { a[q] = ri[k2 + q]; b[q] = rj[k2 + q]; }
#pragma unroll
                        for (int q = 0; q < 8; ++q) s -= a[q] * b[q];
                    }
                    for (; k2 < j; ++k2) s -= ri[k2] * rj[k2];
                }
                const double dj = bcast(s, j);
                if (!(dj > 0.0)) { posdef = false; break; }
                const double inv = 1.0 / sqrt(dj);
                __syncthreads();
                if (lane == j) myinv = inv;
                if (act && li > j) L.M[tri(li, j)] = s * inv;
                __syncthreads();
            }
            QSTAMP(3);
            if (!posdef) { st = 4; break; }
            // two solves with the factor: forward by columns (same subtraction order as the oracle's rows), backward by columns
            auto solve = [&](double rhs) -> double {
                if constexpr (FAST) { const double x = dense40_solve(W, rhs, lz); return act ? x : 0.0; }
                double r = rhs;
                for (int k2 = 0; k2 < n; ++k2) {
                    const double xk = bcast(r, k2) * bcast(myinv, k2);
                    if (lane == k2) r = xk;
                    if (act && li > k2) r -= L.M[tri(li, k2)] * xk;
                }
                for (int k2 = n - 1; k2 >= 0; --k2) {
                    const double xk = bcast(r, k2) * bcast(myinv, k2);
                    if (lane == k2) r = xk;
                    if (act && li < k2) r -= L.M[tri(k2, li)] * xk;
                }
                return r;
            };
            const double da = solve(act ? -rs + (-ll - Dl * rl) - (-lu - Du * ru) : 0.0);
            double dtl = da + rl, dtu = -da + ru;
            double dll = -ll - Dl * dtl, dlu = -lu - Du * dtu;
            auto ratio = [&]() -> double {
                double a = 1.0;
                if (act) {
                    if (dtl < 0) a = fmin(a, -tl / dtl);
                    if (dtu < 0) a = fmin(a, -tu / dtu);
                    if (dll < 0) a = fmin(a, -ll / dll);
                    if (dlu < 0) a = fmin(a, -lu / dlu);
                }
                return wave_min(a);
            };
            double amax = ratio();
            const double muaff = wave_sum(act ? (tl + amax * dtl) * (ll + amax * dll) + (tu + amax * dtu) * (lu + amax * dlu) : 0.0) / (2.0 * n);
            double sigma = muaff / mu; sigma = sigma * sigma * sigma;
            if (alpha_prev < ADMPC_QUAD_IPM_BLOCKED_STEP) sigma = 1.0;
            const double smu = sigma * mu;
            const double cl = act ? (smu - (cons ? 0.0 : dtl * dll)) / tl : 0.0, cu = act ? (smu - (cons ? 0.0 : dtu * dlu)) / tu : 0.0;
            const double d = solve(act ? -rs + (cl - ll - Dl * rl) - (cu - lu - Du * ru) : 0.0);
            dtl = d + rl; dtu = -d + ru;
            dll = cl - ll - Dl * dtl; dlu = cu - lu - Du * dtu;
            amax = ratio();
            double tau = 1.0 - muaff; tau = fmax(tau, 0.995); tau = fmin(tau, 0.999999);
            const double alpha = fmin(tau * amax, 1.0);
            if (act) {
                du += alpha * d;
                tl = fmax(tl + alpha * dtl, 1e-40); tu = fmax(tu + alpha * dtu, 1e-40);
                ll = fmax(ll + alpha * dll, 1e-40); lu = fmax(lu + alpha * dlu, 1e-40);
            }
            alpha_prev = alpha;
            __syncthreads();
            QSTAMP(4);
        }
        // ---- 4. expansion, full step, cost (oracle: rti_step)
        __syncthreads();
        if (FAST) { if (act) L.M[tri(li, li)] = 0.0; }               // (the factor never writes its diagonal; kept for the next instance)
        L.vec[lane] = act ? du : 0.0;
        if (lane < QX) L.xnew[lane] = xb[lane] + (x0[lane] - xb[lane]);
        __syncthreads();
        bool bad = st != 0;
        const double un = ubar_i + du;
        if (act && !(fabs(un) <= 1e300)) bad = true;
        double dx = lane < QX ? x0[lane] - xb[lane] : 0.0;
        double J = 0.0;
        if (act) { const double e = un - yr[ji * QY + QX + mi]; J += 0.5 * Rw * e * e; }
        for (int k = 0; k < N; ++k) {
            if (lane < QX) L.gam[lane] = dx;                         // dx_k of all components
            __syncthreads();
            double dn = 0.0;
            if (lane < QX) {
                dn = L.b[k * QX + lane];
                for (int cc = 0; cc < QX; ++cc) dn += L.A[(k * QX + lane) * QX + cc] * L.gam[cc];
                for (int m = 0; m < QU; ++m) dn += L.B[(k * QX + lane) * QU + m] * L.vec[k * QU + m];
                const double xv = L.xbs[(k + 1) * QX + lane] + dn;
                L.xnew[(k + 1) * QX + lane] = xv;
                if (!(fabs(xv) <= 1e300)) bad = true;
            }
            __syncthreads();
            dx = dn;
        }
        for (int i = lane; i < (N + 1) * QX; i += 64) {
            const int k = i / QX, cc = i - k * QX;
            const double e = L.xnew[i] - L.yrs[k * QY + cc];
            J += 0.5 * (k < N ? Ts * c->W[cc] : c->We[cc]) * e * e;        // (once per instance, parallel over lanes)
        }
        bad = __any(bad) != 0;
        J = wave_sum(J);
        if (!bad) {
            for (int i = lane; i < (N + 1) * QX; i += 64) xb[i] = L.xnew[i];
            if (act) ub[li] = un;
        }
        QSTAMP(5);
        if (lane == 0) {
            if (costg) costg[inst] = bad ? INFINITY : J;
            if (statusg) statusg[inst] = bad ? 4 : 0;
            if (itersg) itersg[inst] = it;
        }
        __syncthreads();
        inst = next_inst(inst);
    }
    QFLUSH();
}


// =====================================================================================================================================
// N = 20 -- the default horizon of the reference class (quad_3d_optimizer.py:28) -- as TWO cooperating waves per instance (round 4): each
// wave shoots, condenses (40 inputs), factorises and expands 10 stages with the machinery of the one-wave kernel above; the two segments
// are coupled through the 13 states at the cut, z = dx_10:  z = Bbar U_0 + c  with multiplier nu.  The bordered factorisation (the 13 rows
// of Bbar ride along in wave 0's lanes 40..52, the 13 rows of Hzu in wave 1's), the Schur blocks on v_mfma_f64_16x16x4_f64 and the four
// solution operators of the cut are seg_cut.h -- the scheme of the car's segmented kernel (admpc_seg.hip; DESIGN.md section 4, kernel S) with
// D = 13.  Same Newton steps as the oracle's dense 80-input box QP (oracle/quad_oracle.c: box_qp) in another elimination order.
// =====================================================================================================================================
#include "seg_cut.h"
#define QSYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#define QWAVE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct QSegLds {      // doubles; per wave.  The linearisation (A, B, b of 10 stages: 18.7 KB) does NOT live here: it waits in the wave's slot of global
    // memory (L2) and passes through a two-stage buffer in the loops that walk the stages (condensing, expansion) -- with it resident the
    // kernel fitted one workgroup per CU, without it two.  gam (the Gamma / Phi components of the current stage, rows of 64 lanes: the z lanes
    // publish too) overlays the factor, which only the interior point writes.
    static constexpr int Ns = 10, n = 40, NB = 14, NR = 54, D = 13, GS = 16;
    static constexpr int STG = 236;                                    // one stage of the linearisation: A [169], B [52], b [13] (+ 2)
    static constexpr int SLOT = Ns * 234;                              // doubles per wave in the slot buffer: A [10][169] | B [10][52] | b [10][13]
    static constexpr int oStg = 0, oH = oStg + 2 * STG, oHb = oH + 820, oL = oHb + NB * n, oLb = oL + 820, oGam = oL,
                         oEx = oLb + NB * n, oVec = oEx + 192, oXnew = oVec + 64, oXbs = oXnew + 144, oYrs = oXbs + 144, oXhs = oYrs + 184,
                         oWts = oXhs + 16, oWv = oWts + 28, seg = oWv + 16;
    static_assert(QX * 64 <= 820 + NB * n, "gam overlays L and Lb");
    // interface block of a segment
    static constexpr int IF_SC = 0, IF_HZZ = IF_SC + NB * NB, IF_C = IF_HZZ + D * GS, IF_ZB = IF_C + 16, IF_Z = IF_ZB + 16, IF_DZ = IF_Z + 16,
                         IF_GZ = IF_DZ + 16, IF_RED = IF_GZ + 16, IF_BU = IF_RED + 32, IFS = IF_BU + 16;
    static constexpr int oIF = 2 * seg, oYM = oIF + 2 * IFS, oPI = oYM + 4 * D * GS, oWG = oPI + D * GS, total = oWG + 8;
};

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void admpc_quad_seg_kernel(const Cfg* __restrict__ c, int B, const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                             const double* __restrict__ yrefeg, double* __restrict__ xbarg, double* __restrict__ ubarg,
                                                             double* __restrict__ costg, int32_t* __restrict__ statusg, int32_t* __restrict__ itersg,
                                                             int* __restrict__ ticket, const double* __restrict__ gpsg, const int32_t* __restrict__ routeg, int which,
                                                             double* __restrict__ slotg)
{
    using LD = QSegLds;
    constexpr int Ns = LD::Ns, n = LD::n, NR = LD::NR, D = LD::D, GS = LD::GS, NT = 2 * Ns;
    extern __shared__ double lds_raw[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool first = wv == 0, last = wv == 1;
    const int k0 = wv * Ns;
    double* const P = lds_raw + wv * LD::seg;
    double* const stg = P + LD::oStg;                                // [2][236] the stage buffer: A_k, B_k, b_k of the stage a loop works on / the next
    double* const slot = slotg + ((size_t)blockIdx.x * 2 + wv) * LD::SLOT;      // this wave's linearisation in global memory (L2), rewritten per instance
    double* const slA = slot; double* const slB = slot + Ns * QX * QX; double* const slb = slB + Ns * QX * QU;
    double* const Hp = P + LD::oH; double* const Hb = P + LD::oHb; double* const Lp = P + LD::oL; double* const Lb = P + LD::oLb;
    double* const gam = P + LD::oGam; double* const vec = P + LD::oVec; double* const xnew = P + LD::oXnew; double* const xbs = P + LD::oXbs;
    double* const yrs = P + LD::oYrs; double* const xhs = P + LD::oXhs; double* const wts = P + LD::oWts; double* const wvec = P + LD::oWv;
    double* const ifb = lds_raw + LD::oIF;
    double* const IFm = ifb + wv * LD::IFS;
    double* const YM = lds_raw + LD::oYM; double* const PI = lds_raw + LD::oPI;
    int* const wgw = reinterpret_cast<int*>(lds_raw + LD::oWG);
    double* const ex = P + LD::oEx;                                  // [192] exchange buffers / pivots of the factorisation (cb, invd, second buffer)
    const Dense40bLds W{Hp, Hb, Lp, Lb, ex, ex + 64};
#define QRED(s_, ph_, i_) ifb[(s_) * LD::IFS + LD::IF_RED + (ph_) * 8 + (i_)]
    const bool act = lane < n;
    const bool zl = !first && lane >= n && lane < n + D;             // z rows (wave 1); wave 0's lanes 40..52 carry the rows of Bbar
    const int li = act ? lane : 0, ji = li / QU, mi = li - ji * QU;
    const int zi = zl ? lane - n : 0;
    const double Ts = c->Ts;
    if (lane < QX) { wts[lane] = Ts * c->W[lane]; wts[QX + lane] = c->We[lane]; }
    const double Rw = Ts * c->W[QX + mi], lbm = c->lbu[mi], ubm = c->ubu[mi];
    const double thr0 = c->ipm_thr0, mu0 = c->ipm_mu0, tolc = c->ipm_tol_comp, tolr = c->ipm_tol_res;
    const int itmax = c->ipm_iter_max;
    const double inv2n = 1.0 / (2.0 * (double)(NT * QU));
    typedef double d4q __attribute__((ext_vector_type(4)));
    // stage k of the linearisation from the slot: 234 doubles, four loads per lane, issued a stage ahead of their use
    auto stage_fetch = [&](const int k, double (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int e = lane + 64 * q; e = e < 234 ? e : 233;
            const double* src = e < 169 ? slA + k * 169 + e : (e < 221 ? slB + k * 52 + (e - 169) : slb + k * 13 + (e - 221));
            t[q] = *src;
        }
    };
    auto stage_commit = [&](double* dst, const double (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = lane + 64 * q; if (e < 234) dst[e] = t[q]; }
    };

    int inst = blockIdx.x;
    for (;;) {
        // ---- the workgroup's next instance (wave 0 draws, both waves take it)
        if (wv == 0 && lane == 0) wgw[0] = inst;
        QSYNC();
        inst = __builtin_amdgcn_readfirstlane(wgw[0]);
        if (inst >= B) break;
        auto next_inst = [&](int cur) -> int {                       // evaluated by both waves alike only for the static stride; the counter is wave 0's
            if (!ticket) return cur + (int)gridDim.x;
            int tk = 0;
            if (wv == 0) { if (lane == 0) tk = atomicAdd(ticket, 1); tk = (int)gridDim.x + __builtin_amdgcn_readfirstlane(tk); }
            return tk;
        };
        if (routeg && routeg[inst] != which) { inst = next_inst(inst); QSYNC(); continue; }
        double* xb = xbarg + (size_t)inst * (NT + 1) * QX;
        double* ub = ubarg + (size_t)inst * NT * QU;
        const double* yr = yrefg + (size_t)inst * NT * QY;
        const double* ye = yrefeg + (size_t)inst * QX;
        const double* x0 = x0g + (size_t)inst * QX;
        const double* gq = gpsg ? gpsg + (size_t)inst * QX : x0;
        double* xbg = xb + (size_t)k0 * QX;                           // the segment's rows
        double* ubg = ub + (size_t)k0 * QU;
        const double* yrg = yr + (size_t)k0 * QY;
        for (int i = lane; i < (Ns + 1) * QX; i += 64) xbs[i] = xbg[i];
        for (int i = lane; i < Ns * QY + QX; i += 64) yrs[i] = i < Ns * QY ? yrg[i] : (last ? ye[i - Ns * QY] : yrg[i]);      // row 10: terminal reference (last) / the next stage's states
        // ---- 1. shooting of the segment's 10 stages: 17 tasks (state + one sensitivity column) per stage
        for (int t = lane; t < Ns * (QX + QU); t += 64) {
            const int k = t / (QX + QU), col = t - k * (QX + QU);
            double x[QX], u[QU], phi[QX], sc[QX];
#pragma unroll
            for (int i = 0; i < QX; ++i) x[i] = xbg[k * QX + i];
#pragma unroll
            for (int m = 0; m < QU; ++m) u[m] = ubg[k * QU + m];
            rk4_col(c, x, u, k + k0 == 0, gq, c->Ts, col, phi, sc);
#pragma unroll
            for (int i = 0; i < QX; ++i) {
                if (col < QX) slA[(k * QX + i) * QX + col] = sc[i]; else slB[(k * QX + i) * QU + (col - QX)] = sc[i];
                if (col == 0) slb[k * QX + i] = phi[i] - xbg[(k + 1) * QX + i];
            }
        }
        // the slot is read back below: the stores have to be out, and the vector L1 may still hold the lines the PREVIOUS instance of this wave
        // read here (it does not follow the wave's own stores): drop them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        {
            double t0_[4];
            stage_fetch(0, t0_);
            stage_commit(stg, t0_);
        }
        QWAVE();
        // ---- 2. condensing: lane i < 40 carries column i of Gamma_k = d x_k / d U_s, lanes 40..52 of wave 1 the columns of Phi_k = d x_k / d z
        double hrow[n];
#pragma unroll
        for (int j = 0; j < n; ++j) hrow[j] = 0.0;
        double g[QX];
#pragma unroll
        for (int i = 0; i < QX; ++i) g[i] = (zl && zi == i) ? 1.0 : 0.0;
        if (lane < QX) xhs[lane] = first ? x0[lane] - xb[lane] : 0.0;
        const double ubar_i = ubg[li];
        double grad = act ? Rw * (ubar_i - yrg[ji * QY + QX + mi]) : 0.0;
        d4q hzz = {0.0, 0.0, 0.0, 0.0};                              // Hzz = sum Phi' Q Phi on one MFMA tile (wave 1)
        const int r16 = lane & 15, kq = lane >> 4;
        QWAVE();
        if (!first) {
            // the segment's first stage is the cut itself: Phi = I, Gamma = 0, free response 0 -- its tracking cost belongs to this segment
            if (zl) grad += wts[zi] * (xbs[zi] - yrs[zi]);
#pragma unroll
            for (int v = 0; v < 4; ++v) { const int i = kq + 4 * v; if (i == r16 && i < QX) hzz[v] = wts[i]; }
        }
        for (int k = 0; k < Ns; ++k) {
            double gn[QX];
            const double* Ak = stg + (k & 1) * LD::STG; const double* Bk = Ak + 169; const double* bk = Ak + 221;
            double tnext[4];
            if (k + 1 < Ns) stage_fetch(k + 1, tnext);                 // lands under this stage's arithmetic
            const int rr = lane < QX ? lane : 0;
            double xn = bk[rr];
#pragma unroll
            for (int cc = 0; cc < QX; ++cc) xn += Ak[rr * QX + cc] * xhs[cc];
            double Ar[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) { const int e = lane + 64 * m; Ar[m] = Ak[e < QX * QX ? e : QX * QX - 1]; }
#pragma unroll
            for (int r = 0; r < QX; ++r) {
                double sacc = 0.0;
#pragma unroll
                for (int cc = 0; cc < QX; ++cc) sacc += bcast(Ar[(r * QX + cc) / 64], (r * QX + cc) % 64) * g[cc];
                gn[r] = zl ? sacc : (act ? (ji == k ? Bk[r * QU + mi] : (ji < k ? sacc : 0.0)) : 0.0);
            }
#pragma unroll
            for (int r = 0; r < QX; ++r) { g[r] = gn[r]; gam[r * 64 + lane] = gn[r]; }
            if (lane < QX) xhs[lane] = xn;
            if (k + 1 < Ns) stage_commit(stg + ((k + 1) & 1) * LD::STG, tnext);
            QWAVE();
            const bool cost_k = k + 1 < Ns || last;                  // stage k0 + 10 is the next segment's; the horizon's last stage carries the terminal weights
            if (cost_k) {
                const double* ref = yrs + (k + 1) * QY;
                const int lim = (k + 1) * QU;
                const int wo = (k + 1 < Ns) ? 0 : QX;
                double wg[QX];
#pragma unroll
                for (int cc = 0; cc < QX; ++cc) {
                    const double wq = wts[wo + cc];
                    wg[cc] = g[cc] * wq;
                    if (wq != 0.0) grad += wg[cc] * (xbs[(k + 1) * QX + cc] + xhs[cc] - ref[cc]);
                }
                double Rb[QX][3];
#pragma unroll
                for (int cc = 0; cc < QX; ++cc)
#pragma unroll
                    for (int m = 0; m < 3; ++m) Rb[cc][m] = gam[cc * 64 + 16 * m + (lane & 15)];
                static_for<0, 10>([&](auto qq) __attribute__((always_inline)) {
                    constexpr int i2 = 4 * decltype(qq)::value;
                    if (i2 < lim) {
                        static_for<0, QX>([&](auto ccc) __attribute__((always_inline)) {
                            constexpr int cc = decltype(ccc)::value;
                            fmac_rowbc4_ld<i2 % 16>(hrow[i2], hrow[i2 + 1], hrow[i2 + 2], hrow[i2 + 3], Rb[cc][i2 / 16], wg[cc]);
                        });
                    }
                });
                if (!first) {
                    // Hzz += Phi' W Phi: four K-steps of a 16 x 16 x 4 tile (13 components, the rest weightless); operand (i = lane & 15, k = lane >> 4)
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const int comp = 4 * t4 + kq;
                        const int cc = comp < QX ? comp : 0;
                        const double ph = gam[cc * 64 + n + (r16 < D ? r16 : 0)];
                        const double wq = comp < QX ? wts[wo + cc] : 0.0;
                        hzz = __builtin_amdgcn_mfma_f64_16x16x4f64(ph * wq, ph, hzz, 0, 0, 0);
                    }
                }
            }
            QWAVE();
        }
        // ---- the packed rows: inputs -> H (+ R on the diagonal), z lanes -> Hzu (border rows 0..12), Hzz; wave 0: Bbar = Gamma at the cut, c
        store_row_40(hrow, lds_byte_addr(Hp + (act ? tri(li, 0) : 0)));
        if (zl) {
#pragma unroll
            for (int j = 0; j < n; ++j) Hb[zi * n + j] = hrow[j];
        }
        if (!first) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { const int i = kq + 4 * v; if (i < D && r16 < D) IFm[LD::IF_HZZ + i * GS + r16] = hzz[v]; }
        }
        if (first && act) {
#pragma unroll
            for (int r = 0; r < QX; ++r) Hb[r * n + lane] = g[r];
        }
        if (act) Hb[13 * n + lane] = 0.0;                             // the fourteenth border row of the 54-row blocks: unused
        if (first && lane < QX) IFm[LD::IF_C + lane] = xhs[lane];
        QWAVE();
        if (act) { Hp[tri(li, li)] += Rw; Lp[tri(li, li)] = 0.0; }
        const double gz0 = zl ? grad : 0.0;                           // d f_1 / d z at (z = 0, U = 0)
        QSYNC();
        if (!first && lane < D) { IFm[LD::IF_Z + lane] = ifb[LD::IF_C + lane]; }      // cold start: z = c_0 (U_0 = 0)
        QWAVE();
        // ---- 3. box QP, the two segments coupled through the cut
        const double lo = lbm - ubar_i, hi = ubm - ubar_i;
        double du = 0.0;
        double tl = act ? fmax(du - lo, thr0) : 1.0, tu = act ? fmax(hi - du, thr0) : 1.0;
        double ll = act ? mu0 / tl : 0.0, lu = act ? mu0 / tu : 0.0;
        double alpha_prev = 1.0;
        int it = 0, st = 0;
        bool cons = false;
        for (;; ++it) {
            int lz = lane;
            asm volatile("" : "+v"(lz));
            // local gradients: inputs  grad + H du (+ Hzu' z) - ll + lu ;  z rows (wave 1)  gz0 + Hzu du + Hzz z
            double hdu = dense40b_symv<NR>(W, lane, act ? du : 0.0, lz);
            if (!first) {
                const double* base = act ? Hb + lane : IFm + LD::IF_HZZ + zi * GS;
                const int stride = act ? n : 1;
                double a = 0.0;
#pragma unroll
                for (int cc = 0; cc < D; ++cc) a = fma(base[cc * stride], IFm[LD::IF_Z + cc], a);
                hdu += a;
            }
            const double rs_loc = act ? grad + hdu - ll + lu : 0.0;
            const double gzc = zl ? gz0 + hdu : 0.0;
            if (zl) IFm[LD::IF_GZ + zi] = gzc;
            QSYNC();
            // the full condensed stationarity residual (the oracle's rs): wave 0's inputs also see the second segment through the cut
            double rs_full = rs_loc;
            if (first && act) {
                double a = 0.0;
#pragma unroll
                for (int r = 0; r < D; ++r) a = fma(Hb[r * n + lane], ifb[LD::IFS + LD::IF_GZ + r], a);
                rs_full += a;
            }
            const double rl = du - lo - tl, ru = hi - du - tu;
            {
                const double mus = wave_sum(act ? tl * ll + tu * lu : 0.0);
                const double cmx = wave_max(act ? fmax(tl * ll, tu * lu) : 0.0);
                const double rmx = wave_max(act ? fmax(fabs(rs_full), fmax(fabs(rl), fabs(ru))) : 0.0);
                if (lane == 0) { QRED(wv, 0, 0) = mus; QRED(wv, 0, 1) = cmx; QRED(wv, 0, 2) = rmx; }
            }
            QSYNC();
            const double mu = (QRED(0, 0, 0) + QRED(1, 0, 0)) * inv2n;
            const double cmax = fmax(QRED(0, 0, 1), QRED(1, 0, 1));
            const double rmax = OpMaxNan::f(QRED(0, 0, 2), QRED(1, 0, 2));
            if (!(mu == mu) || !(rmax == rmax)) { st = 4; break; }
            if ((cmax <= tolc && rmax <= tolr) || it >= itmax + (cons ? ADMPC_QUAD_IPM_FALLBACK_ITER : 0)) break;
            if (!cons && it >= ADMPC_QUAD_IPM_FALLBACK_ITER) {
                cons = true;
                du = 0.0;
                tl = act ? fmax(du - lo, thr0) : 1.0; tu = act ? fmax(hi - du, thr0) : 1.0;
                ll = act ? mu0 / tl : 0.0; lu = act ? mu0 / tu : 0.0;
                alpha_prev = 1.0;
                if (!first && lane < D) IFm[LD::IF_Z + lane] = ifb[LD::IF_C + lane];
                QSYNC();
                --it;
                continue;
            }
            const double Dl = act ? ll / tl : 0.0, Du = act ? lu / tu : 0.0;
            dense40b_factorise<NR>(W, act ? Dl + Du : 1.0, lz);
            dense40b_schur<LD::NB>(W, IFm + LD::IF_SC, LD::NB, lane);
            QSYNC();
            if (wv == 0) cut_operators2<D, GS>(ifb + LD::IF_SC, LD::NB, ifb + LD::IFS + LD::IF_HZZ, GS, 0.0, -1, ifb + LD::IFS + LD::IF_SC, LD::NB, PI, YM, lz);
            // one coupled solve: right-hand side y on the inputs, the reduced stationarity of z on wave 1's z rows
            auto solve = [&](const double yu, const double yz) -> double {
                double y = act ? yu : (zl ? yz : 0.0);
                const unsigned pub = lds_byte_addr(W.cb + (lz & 15));
                fwd_subst_40_b<NR>(y, dense40b_row_addr<NR>(Lp, Lb, lz), pub);
                if (lz >= n && lz < n + 16) IFm[LD::IF_ZB + lz - n] = y;
                QSYNC();
                const double a = cut_apply2<D, GS>(YM, first, ifb + LD::IF_ZB, ifb + LD::IFS + LD::IF_ZB, lz);
                if (lz < 16) wvec[lz] = lz < D ? (first ? -a : a) : 0.0;
                if (!first && lz < D) IFm[LD::IF_DZ + lz] = a;
                QWAVE();
                double x = y * W.invd[act ? lz : 0];
#pragma unroll
                for (int bq = 0; bq < D; ++bq) x = fma(-Lb[bq * n + (act ? lz : 0)], wvec[bq], x);
                bwd_subst_40(x, lds_byte_addr(Lp + (act ? lz : 0)), pub);
                return act ? x : 0.0;
            };
            const double da = solve(-rs_loc + (-ll - Dl * rl) - (-lu - Du * ru), -gzc);
            double dtl = da + rl, dtu = -da + ru;
            double dll = -ll - Dl * dtl, dlu = -lu - Du * dtu;
            auto ratio_local = [&]() -> double {
                double a = 1.0;
                if (act) {
                    if (dtl < 0) a = fmin(a, -tl / dtl);
                    if (dtu < 0) a = fmin(a, -tu / dtu);
                    if (dll < 0) a = fmin(a, -ll / dll);
                    if (dlu < 0) a = fmin(a, -lu / dlu);
                }
                return wave_min(a);
            };
            {
                const double am = ratio_local();
                const double sd = wave_sum(act ? dtl * dll + dtu * dlu : 0.0);
                if (lane == 0) { QRED(wv, 1, 0) = am; QRED(wv, 1, 1) = sd; }
            }
            QSYNC();
            double amax = fmin(QRED(0, 1, 0), QRED(1, 1, 0));
            // sum (t + a dt)(l + a dl) = (1 - a) sum t l + a^2 sum dt dl  (the predictor's right-hand side is t l: t dl + l dt = -t l exactly)
            const double muaff = ((1.0 - amax) * mu / inv2n + amax * amax * (QRED(0, 1, 1) + QRED(1, 1, 1))) * inv2n;
            double sigma = muaff / mu; sigma = sigma * sigma * sigma;
            if (alpha_prev < ADMPC_QUAD_IPM_BLOCKED_STEP) sigma = 1.0;
            const double smu = sigma * mu;
            const double cl = act ? (smu - (cons ? 0.0 : dtl * dll)) / tl : 0.0, cu = act ? (smu - (cons ? 0.0 : dtu * dlu)) / tu : 0.0;
            const double d = solve(-rs_loc + (cl - ll - Dl * rl) - (cu - lu - Du * ru), -gzc);
            dtl = d + rl; dtu = -d + ru;
            dll = cl - ll - Dl * dtl; dlu = cu - lu - Du * dtu;
            {
                const double am = ratio_local();
                if (lane == 0) QRED(wv, 2, 0) = am;
            }
            QSYNC();
            amax = fmin(QRED(0, 2, 0), QRED(1, 2, 0));
            double tau = 1.0 - muaff; tau = fmax(tau, 0.995); tau = fmin(tau, 0.999999);
            const double alpha = fmin(tau * amax, 1.0);
            if (act) {
                du += alpha * d;
                tl = fmax(tl + alpha * dtl, 1e-40); tu = fmax(tu + alpha * dtu, 1e-40);
                ll = fmax(ll + alpha * dll, 1e-40); lu = fmax(lu + alpha * dlu, 1e-40);
            }
            if (!first && lane < D) IFm[LD::IF_Z + lane] += alpha * IFm[LD::IF_DZ + lane];
            alpha_prev = alpha;
            QWAVE();
        }
        // ---- 4. expansion from the cut state of the returned inputs, full step, cost
        if (first) {
            const double duv = act ? du : 0.0;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                const double v = wave_sum(Hb[r * n + li] * duv);
                if (lane == 0) IFm[LD::IF_BU + r] = v + IFm[LD::IF_C + r];
            }
        }
        QSYNC();
        vec[lane] = act ? du : 0.0;
        bool bad = st != 0;
        const double un = ubar_i + du;
        if (act && !(fabs(un) <= 1e300)) bad = true;
        double dx = lane < QX ? (first ? x0[lane] - xb[lane] : ifb[LD::IF_BU + lane]) : 0.0;
        if (lane < QX) xnew[lane] = xbs[lane] + dx;
        double J = 0.0;
        if (act) { const double e = un - yrg[ji * QY + QX + mi]; J += 0.5 * Rw * e * e; }
        {
            double t0_[4];
            stage_fetch(0, t0_);
            stage_commit(stg, t0_);
        }
        QWAVE();
        for (int k = 0; k < Ns; ++k) {
            const double* Ak = stg + (k & 1) * LD::STG; const double* Bk = Ak + 169; const double* bk = Ak + 221;
            double tnext[4];
            if (k + 1 < Ns) stage_fetch(k + 1, tnext);
            if (lane < QX) ex[lane] = dx;
            QWAVE();
            double dn = 0.0;
            if (lane < QX) {
                dn = bk[lane];
                for (int cc = 0; cc < QX; ++cc) dn += Ak[lane * QX + cc] * ex[cc];
                for (int m = 0; m < QU; ++m) dn += Bk[lane * QU + m] * vec[k * QU + m];
                const double xv = xbs[(k + 1) * QX + lane] + dn;
                xnew[(k + 1) * QX + lane] = xv;
                if (!(fabs(xv) <= 1e300)) bad = true;
            }
            if (k + 1 < Ns) stage_commit(stg + ((k + 1) & 1) * LD::STG, tnext);
            QWAVE();
            dx = dn;
        }
        // tracking cost of the segment's stages: k0 .. k0 + 9 with the stage weights; the last segment adds the terminal stage
        for (int i = lane; i < (Ns + 1) * QX; i += 64) {
            const int k = i / QX, cc = i - k * QX;
            if (k < Ns || last) {
                const double e = xnew[i] - yrs[k * QY + cc];
                J += 0.5 * (k < Ns ? Ts * c->W[cc] : c->We[cc]) * e * e;
            }
        }
        const bool anyb = __any(bad) != 0;
        J = wave_sum(J);
        if (lane == 0) { QRED(wv, 3, 0) = J; QRED(wv, 3, 1) = anyb ? 1.0 : 0.0; }
        QSYNC();                                                       // every wave has read what it needs of xbar (the row at the cut is shared)
        const bool badall = QRED(0, 3, 1) != 0.0 || QRED(1, 3, 1) != 0.0;
        if (!badall) {
            for (int i = lane + (first ? 0 : QX); i < (Ns + 1) * QX; i += 64) xbg[i] = xnew[i];      // rows k0 + 1 .. k0 + 10 (wave 0: row 0 as well)
            if (act) ubg[li] = un;
        }
        if (wv == 0 && lane == 0) {
            if (costg) costg[inst] = badall ? INFINITY : QRED(0, 3, 0) + QRED(1, 3, 0);
            if (statusg) statusg[inst] = badall ? 4 : 0;
            if (itersg) itersg[inst] = it;
        }
        inst = next_inst(inst);
        QSYNC();
    }
#undef QRED
}

// ---- horizons beyond 16 (N nu up to 96 inputs; the reference class defaults to n_nodes = 20, quad_3d_optimizer.py:29): the same
// algorithm with one THREAD per input in a workgroup of two wavefronts.  What the one-wave kernel exchanges inside the wave (v_readlane
// broadcasts, DPP reductions) goes through LDS and workgroup barriers here -- about a thousand barriers per interior-point iteration:
// a compatibility path (the shipped horizon N = 10 and everything up to 16 stay on the kernel above), with the oracle's summation
// order in the factorisation and the substitutions, so that iteration counts and results follow the oracle as on the other paths.
#define QW_NT 128
struct WideRed { double* red; double* slot; };
__device__ __forceinline__ double qw_sum(double v, const WideRed& R, int tid) {
    v = wave_sum(v); if ((tid & 63) == 0) R.red[tid >> 6] = v; __syncthreads(); const double r = R.red[0] + R.red[1]; __syncthreads(); return r;
}
__device__ __forceinline__ double qw_max(double v, const WideRed& R, int tid) {
    v = wave_max(v); if ((tid & 63) == 0) R.red[tid >> 6] = v; __syncthreads(); const double r = OpMaxNan::f(R.red[0], R.red[1]); __syncthreads(); return r;
}
__device__ __forceinline__ double qw_min(double v, const WideRed& R, int tid) {
    v = wave_min(v); if ((tid & 63) == 0) R.red[tid >> 6] = v; __syncthreads(); const double r = fmin(R.red[0], R.red[1]); __syncthreads(); return r;
}

__global__ __launch_bounds__(QW_NT) void admpc_quad_solve_wide_kernel(const Cfg* __restrict__ c, int B, const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                                      const double* __restrict__ yrefeg, double* __restrict__ xbarg, double* __restrict__ ubarg,
                                                                      double* __restrict__ costg, int32_t* __restrict__ statusg, int32_t* __restrict__ itersg,
                                                                      int* __restrict__ ticket, const double* __restrict__ gpsg, const int32_t* __restrict__ routeg, int which)
{
    extern __shared__ double lds_raw[];
    const int N = c->N, n = N * QU, tid = threadIdx.x;
    Lds L(lds_raw, N);
    double* const invs = lds_raw + quad_lds_doubles(N);             // [n] reciprocal pivots, then [2] broadcast slots, [2] reduction slots
    const WideRed R{ invs + n + 2, invs + n };
    const bool act = tid < n;
    const int li = act ? tid : 0, ji = li / QU, mi = li - ji * QU;
    const double Ts = c->Ts;
    if (tid < QX) { L.wts[tid] = Ts * c->W[tid]; L.wts[QX + tid] = c->We[tid]; }
    const double Rw = Ts * c->W[QX + mi], lbm = c->lbu[mi], ubm = c->ubu[mi];
    const double thr0 = c->ipm_thr0, mu0 = c->ipm_mu0, tolc = c->ipm_tol_comp, tolr = c->ipm_tol_res;
    const int itmax = c->ipm_iter_max;
    auto next_inst = [&](int inst) -> int {                          // block-uniform: the draw of thread 0 through an LDS slot
        if (!ticket) return inst + (int)gridDim.x;
        if (tid == 0) R.red[0] = (double)atomicAdd(ticket, 1);
        __syncthreads();
        const int nx = (int)gridDim.x + (int)R.red[0];
        __syncthreads();
        return nx;
    };
    for (int inst = blockIdx.x; inst < B;) {
        if (routeg && routeg[inst] != which) { inst = next_inst(inst); continue; }      // routed solve: another cluster's instance
        double* xb = xbarg + (size_t)inst * (N + 1) * QX;
        double* ub = ubarg + (size_t)inst * N * QU;
        const double* yr = yrefg + (size_t)inst * N * QY;
        const double* ye = yrefeg + (size_t)inst * QX;
        const double* x0 = x0g + (size_t)inst * QX;
        const double* gq = gpsg ? gpsg + (size_t)inst * QX : x0;
        for (int i = tid; i < (N + 1) * QX; i += QW_NT) L.xbs[i] = xb[i];
        for (int i = tid; i < N * QY + QX; i += QW_NT) L.yrs[i] = i < N * QY ? yr[i] : ye[i - N * QY];
        shoot_instance(c, xb, ub, gq, L, tid, nullptr, QW_NT);
        // ---- condensing (oracle: condense)
        for (int j = 0; j <= li; ++j) if (act) L.H[tri(li, j)] = 0.0;
        double g[QX];
#pragma unroll
        for (int i = 0; i < QX; ++i) g[i] = 0.0;
        if (tid < QX) L.xhs[tid] = x0[tid] - xb[tid];
        const double ubar_i = ub[li];
        double grad = Rw * (ubar_i - yr[ji * QY + QX + mi]);
        __syncthreads();
        for (int k = 0; k < N; ++k) {
            double gn[QX];
            const int rr = tid < QX ? tid : 0;
            double xn = L.b[k * QX + rr];
#pragma unroll
            for (int cc = 0; cc < QX; ++cc) xn += L.A[(k * QX + rr) * QX + cc] * L.xhs[cc];
#pragma unroll
            for (int r = 0; r < QX; ++r) {
                double s = 0.0;
#pragma unroll
                for (int cc = 0; cc < QX; ++cc) s += L.A[(k * QX + r) * QX + cc] * g[cc];
                gn[r] = ji == k ? L.B[(k * QX + r) * QU + mi] : (ji < k ? s : 0.0);
            }
            __syncthreads();                                                     // every read of xhs above is done
#pragma unroll
            for (int r = 0; r < QX; ++r) { g[r] = gn[r]; if (act) L.gam[r * n + tid] = gn[r]; }
            if (tid < QX) L.xhs[tid] = xn;
            __syncthreads();
            const double* ref = L.yrs + (k + 1) * QY;
            const int lim = (k + 1) * QU;
            double wg[QX];
#pragma unroll
            for (int cc = 0; cc < QX; ++cc) {
                const double wq = L.wts[(k + 1 < N ? 0 : QX) + cc];
                wg[cc] = g[cc] * wq;
                if (wq != 0.0) grad += wg[cc] * (L.xbs[(k + 1) * QX + cc] + L.xhs[cc] - ref[cc]);
            }
            if (act && li < lim) {
                for (int j = 0; j <= li; ++j) {
                    double s = L.H[tri(li, j)];
#pragma unroll
                    for (int cc = 0; cc < QX; ++cc) s += wg[cc] * L.gam[cc * n + j];
                    L.H[tri(li, j)] = s;
                }
            }
            __syncthreads();
        }
        if (act) L.H[tri(li, li)] += Rw;
        __syncthreads();
        // ---- box QP (oracle: box_qp)
        const double lo = lbm - ubar_i, hi = ubm - ubar_i;
        double du = 0.0;
        double tl = act ? fmax(du - lo, thr0) : 1.0, tu = act ? fmax(hi - du, thr0) : 1.0;
        double ll = act ? mu0 / tl : 0.0, lu = act ? mu0 / tu : 0.0;
        double alpha_prev = 1.0;
        int it = 0, st = 0;
        bool cons = false;
        for (;; ++it) {
            double rs = grad - ll + lu;
            L.gam[tid] = du;                                                     // (gam is free after the condensing: n <= 96 values here)
            __syncthreads();
            if (act) for (int j = 0; j < n; ++j) rs += L.H[sym(li, j)] * L.gam[j];
            const double rl = du - lo - tl, ru = hi - du - tu;
            const double mu = qw_sum(act ? tl * ll + tu * lu : 0.0, R, tid) / (2.0 * n);
            const double cmax = qw_max(act ? fmax(tl * ll, tu * lu) : 0.0, R, tid);
            const double rmax = qw_max(act ? fmax(fabs(rs), fmax(fabs(rl), fabs(ru))) : 0.0, R, tid);
            if (!(mu == mu) || !(rmax == rmax)) { st = 4; break; }
            if ((cmax <= tolc && rmax <= tolr) || it >= itmax + (cons ? ADMPC_QUAD_IPM_FALLBACK_ITER : 0)) break;
            if (!cons && it >= ADMPC_QUAD_IPM_FALLBACK_ITER) {
                cons = true;
                du = 0.0;
                tl = act ? fmax(du - lo, thr0) : 1.0; tu = act ? fmax(hi - du, thr0) : 1.0;
                ll = act ? mu0 / tl : 0.0; lu = act ? mu0 / tu : 0.0;
                alpha_prev = 1.0;
                --it;
                continue;
            }
            const double Dl = act ? ll / tl : 0.0, Du = act ? lu / tu : 0.0;
            if (act) { for (int j = 0; j <= li; ++j) L.M[tri(li, j)] = L.H[tri(li, j)]; L.M[tri(li, li)] += Dl + Du; }
            __syncthreads();
            // Cholesky M = L L' (lower), row i on thread i, columns left to right (oracle: chol)
            bool posdef = true;
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                if (act && li >= j) {
                    const double* ri = L.M + tri(li, 0);
                    const double* rj = L.M + tri(j, 0);
                    s = ri[j];
                    for (int k2 = 0; k2 < j; ++k2) s -= ri[k2] * rj[k2];
                }
                if (tid == j) R.slot[j & 1] = s;
                __syncthreads();
                const double dj = R.slot[j & 1];
                if (!(dj > 0.0)) { posdef = false; break; }                       // uniform: every thread reads the same slot
                const double inv = 1.0 / sqrt(dj);
                if (tid == j) invs[j] = inv;
                if (act && li > j) L.M[tri(li, j)] = s * inv;
                __syncthreads();
            }
            if (!posdef) { st = 4; break; }
            auto solve = [&](double rhs) -> double {
                double r = rhs;
                for (int k2 = 0; k2 < n; ++k2) {
                    if (tid == k2) { r = r * invs[k2]; R.slot[k2 & 1] = r; }
                    __syncthreads();
                    const double xk = R.slot[k2 & 1];
                    if (act && li > k2) r -= L.M[tri(li, k2)] * xk;
                }
                __syncthreads();
                for (int k2 = n - 1; k2 >= 0; --k2) {
                    if (tid == k2) { r = r * invs[k2]; R.slot[k2 & 1] = r; }
                    __syncthreads();
                    const double xk = R.slot[k2 & 1];
                    if (act && li < k2) r -= L.M[tri(k2, li)] * xk;
                }
                __syncthreads();
                return r;
            };
            const double da = solve(act ? -rs + (-ll - Dl * rl) - (-lu - Du * ru) : 0.0);
            double dtl = da + rl, dtu = -da + ru;
            double dll = -ll - Dl * dtl, dlu = -lu - Du * dtu;
            auto ratio = [&]() -> double {
                double a = 1.0;
                if (act) {
                    if (dtl < 0) a = fmin(a, -tl / dtl);
                    if (dtu < 0) a = fmin(a, -tu / dtu);
                    if (dll < 0) a = fmin(a, -ll / dll);
                    if (dlu < 0) a = fmin(a, -lu / dlu);
                }
                return qw_min(a, R, tid);
            };
            double amax = ratio();
            const double muaff = qw_sum(act ? (tl + amax * dtl) * (ll + amax * dll) + (tu + amax * dtu) * (lu + amax * dlu) : 0.0, R, tid) / (2.0 * n);
            double sigma = muaff / mu; sigma = sigma * sigma * sigma;
            if (alpha_prev < ADMPC_QUAD_IPM_BLOCKED_STEP) sigma = 1.0;
            const double smu = sigma * mu;
            const double cl = act ? (smu - (cons ? 0.0 : dtl * dll)) / tl : 0.0, cu = act ? (smu - (cons ? 0.0 : dtu * dlu)) / tu : 0.0;
            const double d = solve(act ? -rs + (cl - ll - Dl * rl) - (cu - lu - Du * ru) : 0.0);
            dtl = d + rl; dtu = -d + ru;
            dll = cl - ll - Dl * dtl; dlu = cu - lu - Du * dtu;
            amax = ratio();
            double tau = 1.0 - muaff; tau = fmax(tau, 0.995); tau = fmin(tau, 0.999999);
            const double alpha = fmin(tau * amax, 1.0);
            if (act) {
                du += alpha * d;
                tl = fmax(tl + alpha * dtl, 1e-40); tu = fmax(tu + alpha * dtu, 1e-40);
                ll = fmax(ll + alpha * dll, 1e-40); lu = fmax(lu + alpha * dlu, 1e-40);
            }
            alpha_prev = alpha;
            __syncthreads();
        }
        // ---- expansion, full step, cost (oracle: rti_step)
        __syncthreads();
        L.gam[QX + tid] = act ? du : 0.0;                                        // du of all inputs behind the dx slot
        if (tid < QX) L.xnew[tid] = xb[tid] + (x0[tid] - xb[tid]);
        __syncthreads();
        bool bad = st != 0;
        const double un = ubar_i + du;
        if (act && !(fabs(un) <= 1e300)) bad = true;
        double dx = tid < QX ? x0[tid] - xb[tid] : 0.0;
        double J = 0.0;
        if (act) { const double e = un - yr[ji * QY + QX + mi]; J += 0.5 * Rw * e * e; }
        for (int k = 0; k < N; ++k) {
            if (tid < QX) L.gam[tid] = dx;
            __syncthreads();
            double dn = 0.0;
            if (tid < QX) {
                dn = L.b[k * QX + tid];
                for (int cc = 0; cc < QX; ++cc) dn += L.A[(k * QX + tid) * QX + cc] * L.gam[cc];
                for (int m = 0; m < QU; ++m) dn += L.B[(k * QX + tid) * QU + m] * L.gam[QX + k * QU + m];
                const double xv = L.xbs[(k + 1) * QX + tid] + dn;
                L.xnew[(k + 1) * QX + tid] = xv;
                if (!(fabs(xv) <= 1e300)) bad = true;
            }
            __syncthreads();
            dx = dn;
        }
        for (int i = tid; i < (N + 1) * QX; i += QW_NT) {
            const int k = i / QX, cc = i - k * QX;
            const double e = L.xnew[i] - L.yrs[k * QY + cc];
            J += 0.5 * (k < N ? Ts * c->W[cc] : c->We[cc]) * e * e;
        }
        bad = qw_max(bad ? 1.0 : 0.0, R, tid) > 0.0;
        J = qw_sum(J, R, tid);
        if (!bad) {
            for (int i = tid; i < (N + 1) * QX; i += QW_NT) xb[i] = L.xnew[i];
            if (act) ub[li] = un;
        }
        if (tid == 0) {
            if (costg) costg[inst] = bad ? INFINITY : J;
            if (statusg) statusg[inst] = bad ? 4 : 0;
            if (itersg) itersg[inst] = it;
        }
        __syncthreads();
        inst = next_inst(inst);
    }
}

// Cluster of an instance (the reference keeps one acados solver per GP cluster, quad_3d_optimizer.py:207, and picks one per solve from
// the reference state: set_reference_state / set_reference_trajectory :446-452, :485-491 -> gp.py:738-770 select_gp): nearest centroid in
// the selected features of z = [x with the velocity in the BODY frame; u], Euclidean distance, ties to the lower index (numpy.argmin).
// x_sel [B][13] (world-frame velocity: the rotation happens here), u_sel [B][4].  One thread per instance.
__global__ void admpc_quad_select_cluster_kernel(int B, int d, int f0, int f1, int f2, const double* __restrict__ xs, const double* __restrict__ us,
                                                 int K, const double* __restrict__ cent, int32_t* __restrict__ route)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* x = xs + (size_t)b * QX;
    double R[3][3];
    quad_rot(x[3], x[4], x[5], x[6], R);
    double zz[QY];
#pragma unroll
    for (int i = 0; i < QX; ++i) zz[i] = x[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) zz[7 + i] = R[0][i] * x[7] + R[1][i] * x[8] + R[2][i] * x[9];        // v_b = R' v
#pragma unroll
    for (int m = 0; m < QU; ++m) zz[QX + m] = us[(size_t)b * QU + m];
    const int f[3] = { f0, f1, f2 };
    double z[3] = { 0, 0, 0 };
    for (int j = 0; j < d; ++j) {
        double v = 0;
#pragma unroll
        for (int i = 0; i < QY; ++i) v = f[j] == i ? zz[i] : v;
        z[j] = v;
    }
    double best = INFINITY; int bi = 0;
    for (int c = 0; c < K; ++c) {
        double acc = 0.0;
        for (int j = 0; j < d; ++j) { const double e = z[j] - cent[c * d + j]; acc = acc + e * e; }
        const double dist = __dsqrt_rn(acc);
        if (dist < best) { best = dist; bi = c; }
    }
    route[b] = bi;
}
// routed solve: an instance whose route names no cluster fails (status 4, infinite cost) with its iterate untouched
__global__ void admpc_quad_route_invalid_kernel(int B, const int32_t* __restrict__ route, int K, int32_t* __restrict__ status, double* __restrict__ cost,
                                                int32_t* __restrict__ iters)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (route[b] >= 0 && route[b] < K) return;
    if (status) status[b] = 4;
    if (cost) cost[b] = INFINITY;
    if (iters) iters[b] = 0;
}

}  // namespace

// ---- solver_type "SQP" (create_ros_gp_mpc.py:63-68, quad_3d_optimizer.py:203): cfg.sqp_iters QPs per call with acados' stopping test in front of
// every QP but the first.  The test (one thread per instance: the mode serves point references, a handful of instances) follows
// oracle/quad_oracle.c:quad_nlp_residuals -- the multipliers of the last QP by the adjoint recursion of its stationarity (full condensing returns
// none for the dynamics), on the linearisation (Ap, Bp) that QP was built on; the residuals on the linearisation (phi, An, Bn) of the iterate.
//   act [B]: 0 = still iterating, 1 = finished (fst [B] holds its final status)
__global__ void admpc_quad_sqp_test_kernel(const Cfg* __restrict__ c, int B, const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                           const double* __restrict__ yrefeg, const double* __restrict__ xbarg, const double* __restrict__ ubarg,
                                           const double* __restrict__ Apg, const double* __restrict__ Bpg, const double* __restrict__ phig,
                                           const double* __restrict__ Ang, const double* __restrict__ Bng, int32_t* __restrict__ act, int32_t* __restrict__ fst)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || act[b] != 0) return;
    const int N = c->N;
    const double Ts = c->Ts, tol = c->sqp_tol;
    const double* xb = xbarg + (size_t)b * (N + 1) * QX; const double* ub = ubarg + (size_t)b * N * QU;
    const double* yr = yrefg + (size_t)b * N * QY; const double* ye = yrefeg + (size_t)b * QX;
    double rs = 0.0, re = 0.0, ri = 0.0, rc = 0.0;
    auto upn = [](double& acc, double v) { const double a = fabs(v); if (a > acc || !(a == a)) acc = a; };
    for (int i = 0; i < QX; ++i) upn(re, xb[i] - x0g[(size_t)b * QX + i]);
    double pk[QX];
    for (int i = 0; i < QX; ++i) pk[i] = c->We[i] * (xb[N * QX + i] - ye[i]);      // pi_{N-1}; the terminal stationarity row is zero by this definition
    for (int k = N - 1; k >= 0; --k) {
        const double* Ap = Apg + ((size_t)b * N + k) * QX * QX; const double* An = Ang + ((size_t)b * N + k) * QX * QX;
        const double* Bp = Bpg + ((size_t)b * N + k) * QX * QU; const double* Bn = Bng + ((size_t)b * N + k) * QX * QU;
        const double* ph = phig + ((size_t)b * N + k) * QX;
        for (int i = 0; i < QX; ++i) upn(re, ph[i] - xb[(k + 1) * QX + i]);
        for (int m = 0; m < QU; ++m) {
            const double u = ub[k * QU + m];
            const double gr = Ts * c->W[QX + m] * (u - yr[k * QY + QX + m]);
            double mk = gr, a = gr;
            for (int i = 0; i < QX; ++i) { mk += Bp[i * QU + m] * pk[i]; a += Bn[i * QU + m] * pk[i]; }
            upn(rs, a - mk);
            const double vl = c->lbu[m] - u, vu = u - c->ubu[m];
            upn(ri, vl > 0.0 ? vl : 0.0); upn(ri, vu > 0.0 ? vu : 0.0);
            const double ml = mk > 0.0 ? mk : 0.0, mu_ = mk < 0.0 ? -mk : 0.0;
            upn(rc, ml * (u - c->lbu[m])); upn(rc, mu_ * (c->ubu[m] - u));
        }
        if (k >= 1) {
            double pn[QX];
            for (int j = 0; j < QX; ++j) {
                const double gr = Ts * c->W[j] * (xb[k * QX + j] - yr[k * QY + j]);
                double pm = gr, a = gr;
                for (int i = 0; i < QX; ++i) { pm += Ap[i * QX + j] * pk[i]; a += An[i * QX + j] * pk[i]; }
                upn(rs, a - pm);
                pn[j] = pm;
            }
            for (int j = 0; j < QX; ++j) pk[j] = pn[j];
        }
    }
    if (rs <= tol && re <= tol && ri <= tol && rc <= tol) { act[b] = 1; fst[b] = 0; }
}
// behind a QP of an SQP solve: an instance whose QP failed is finished with that status
__global__ void admpc_quad_sqp_merge_kernel(int B, const int32_t* __restrict__ status, int32_t* __restrict__ act, int32_t* __restrict__ fst)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && act[b] == 0 && status[b] != 0) { act[b] = 1; fst[b] = status[b]; }
}
// final statuses: finished instances carry theirs; an instance still iterating at the limit is ACADOS_MAXITER (2) when a tolerance was asked for
__global__ void admpc_quad_sqp_final_kernel(int B, int tol_on, const int32_t* __restrict__ act, const int32_t* __restrict__ fst, int32_t* __restrict__ status)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) status[b] = act[b] ? fst[b] : (tol_on ? 2 : status[b]);
}

struct AdmpcQuadSolver {
    AdmpcQuadConfig cfg;
    AdmpcQuadConfig* d_cfg;
    int* d_ticket;           // work counter of the solve kernel
    int device, num_cu, lds_bytes;
    int generic;             // ADMPC_QUAD_GENERIC=1: the LDS-resident Cholesky path also at N nu = 40 (A/B tests)
    double* d_slot;          // segmented N = 20 kernel: the linearisation of every resident wave (QSegLds::SLOT doubles each), allocated at its first launch
    int wide20;              // ADMPC_QUAD_WIDE=1: N = 20 on the two-wave dense kernel of round 3 instead of the segmented kernel (A/B tests)
    // SQP mode (cfg.sqp_iters > 1), allocated at its first solve: two linearisations (the last QP's and the iterate's), activity flags
    int cap_sqp;
    double *d_A[2], *d_B[2], *d_phi;
    int32_t *d_act, *d_fst, *d_st;
};

namespace {
struct QGuard {
    int prev; bool switched, good;
    explicit QGuard(int dev) : prev(-1), switched(false), good(true) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { good = hipSetDevice(dev) == hipSuccess; switched = good; }
    }
    ~QGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};
}

extern "C" {

void admpc_quad_default_config(AdmpcQuadConfig* c)
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->N = 10; c->ipm_iter_max = 50; c->Ts = 0.1;
    const double w[QY] = { 10, 10, 10, 0, 0.1, 0.1, 0.1, 0.05, 0.05, 0.05, 0.05, 0.05, 0.05, 0.1, 0.1, 0.1, 0.1 };
    for (int i = 0; i < QY; ++i) c->W[i] = w[i];
    for (int m = 0; m < QU; ++m) { c->lbu[m] = 0.0; c->ubu[m] = 1.0; }
    c->mass = 1.0; c->J[0] = 0.03; c->J[1] = 0.03; c->J[2] = 0.06; c->max_thrust = 20.0; c->g = 9.81;
    const double h = cos(M_PI / 4) * (0.47 / 2);
    const double xf[4] = { h, -h, -h, h }, yf[4] = { -h, -h, h, h }, zt[4] = { -0.013, 0.013, -0.013, 0.013 };
    for (int i = 0; i < 4; ++i) { c->x_f[i] = xf[i]; c->y_f[i] = yf[i]; c->z_l_tau[i] = zt[i]; }
    c->ipm_mu0 = 1.0; c->ipm_thr0 = 0.1; c->ipm_tol_comp = 1e-8; c->ipm_tol_res = 1e-8;      // HPIPM mode BALANCE, the reference's setting
    c->sqp_iters = 1; c->sqp_tol = 0.0;                                                       // SQP_RTI, the shipped solver_type
}

int admpc_quad_create(const AdmpcQuadConfig* cfg, int device, AdmpcQuadSolver** out)
{
    if (!cfg || !out) return admpc_set_error(ADMPC_EINVAL, "admpc_quad_create: null argument");
    if (cfg->N < 2 || cfg->N > ADMPC_QUAD_MAX_N) return admpc_set_error(ADMPC_EINVAL, "quad: N must be in [2, 24]");
    if (!(cfg->Ts > 0) || !(cfg->mass > 0) || !(cfg->J[0] > 0 && cfg->J[1] > 0 && cfg->J[2] > 0)) return admpc_set_error(ADMPC_EINVAL, "quad: Ts, mass, J must be positive");
    for (int m = 0; m < QU; ++m) {
        if (!(cfg->W[QX + m] > 0)) return admpc_set_error(ADMPC_EINVAL, "quad: input weights must be positive (strict convexity)");
        if (!(cfg->lbu[m] < cfg->ubu[m])) return admpc_set_error(ADMPC_EINVAL, "quad: lbu < ubu required");
    }
    if (cfg->ipm_iter_max < 1 || !(cfg->ipm_mu0 > 0) || !(cfg->ipm_thr0 > 0)) return admpc_set_error(ADMPC_EINVAL, "quad: bad interior-point parameters");
    if (cfg->sqp_iters < 0 || cfg->sqp_iters > 10000 || !(cfg->sqp_tol >= 0)) return admpc_set_error(ADMPC_EINVAL, "quad: sqp_iters must be in [0, 10000], sqp_tol >= 0");
    if (cfg->n_gp < 0 || cfg->n_gp > ADMPC_QUAD_GP_MAX) return admpc_set_error(ADMPC_EINVAL, "quad: n_gp out of range");
    for (int g = 0; g < cfg->n_gp; ++g) {
        const AdmpcGp& gp = cfg->gp[g];
        bool okf = gp.n_feat >= 1 && gp.n_feat <= ADMPC_GP_MAX_FEAT;
        for (int d = 0; okf && d < gp.n_feat; ++d) okf = gp.feat[d] >= 7 && gp.feat[d] < QY;
        if (gp.out < 7 || gp.out > 9 || !okf || gp.n_points < 0 || gp.n_points > ADMPC_GP_MAX_POINTS)
            return admpc_set_error(ADMPC_EINVAL, "quad GP: out must be in {7,8,9}, 1..3 features in [7,17), n_points <= 32");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return admpc_set_error(ADMPC_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return admpc_set_error(ADMPC_ENODEV, "no such device");
    QGuard guard(device);
    if (!guard.good) return admpc_set_error(ADMPC_EHIP, "hipSetDevice failed");
    AdmpcQuadSolver* s = new (std::nothrow) AdmpcQuadSolver();
    if (!s) return admpc_set_error(ADMPC_ENOMEM, "out of host memory");
    s->cfg = *cfg; s->device = device; s->d_cfg = nullptr; s->d_ticket = nullptr;
    s->d_slot = nullptr;
    s->cap_sqp = 0; s->d_A[0] = s->d_A[1] = s->d_B[0] = s->d_B[1] = s->d_phi = nullptr; s->d_act = s->d_fst = s->d_st = nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete s; return admpc_set_error(ADMPC_EHIP, "hipGetDeviceProperties failed"); }
    s->num_cu = prop.multiProcessorCount;
    s->lds_bytes = (quad_lds_doubles(cfg->N) + (cfg->N * QU > 64 ? cfg->N * QU + 4 : 0)) * (int)sizeof(double);      // wide path: + pivots, broadcast and reduction slots
    { const char* e = getenv("ADMPC_QUAD_GENERIC"); s->generic = e && e[0] == '1'; }
    { const char* e = getenv("ADMPC_QUAD_WIDE"); s->wide20 = e && e[0] == '1'; }
    if (hipMalloc((void**)&s->d_cfg, sizeof(AdmpcQuadConfig)) != hipSuccess || hipMalloc((void**)&s->d_ticket, sizeof(int)) != hipSuccess ||
        hipMemcpy(s->d_cfg, cfg, sizeof(AdmpcQuadConfig), hipMemcpyHostToDevice) != hipSuccess) {
        if (s->d_cfg) (void)hipFree(s->d_cfg);
        if (s->d_ticket) (void)hipFree(s->d_ticket);
        delete s; return admpc_set_error(ADMPC_EHIP, "device allocation failed");
    }
    (void)hipFuncSetAttribute((const void*)admpc_quad_solve_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)admpc_quad_solve_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)admpc_quad_shoot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)admpc_quad_solve_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)admpc_quad_seg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *out = s;
    return ADMPC_OK;
}

void admpc_quad_destroy(AdmpcQuadSolver* s)
{
    if (!s) return;
    QGuard guard(s->device);
    if (s->d_cfg) (void)hipFree(s->d_cfg);
    if (s->d_ticket) (void)hipFree(s->d_ticket);
    if (s->d_slot) (void)hipFree(s->d_slot);
    for (int i = 0; i < 2; ++i) { if (s->d_A[i]) (void)hipFree(s->d_A[i]); if (s->d_B[i]) (void)hipFree(s->d_B[i]); }
    if (s->d_phi) (void)hipFree(s->d_phi);
    if (s->d_act) (void)hipFree(s->d_act);
    if (s->d_fst) (void)hipFree(s->d_fst);
    if (s->d_st) (void)hipFree(s->d_st);
    delete s;
}

static int quad_solve(AdmpcQuadSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* gp_state,
                      double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, const int32_t* route, int which, hipStream_t st)
{
    const bool seg20 = s->cfg.N == 20 && !s->wide20;      // the class default horizon: two cooperating waves per instance, 10 condensed stages each
    const int ldsb = seg20 ? QSegLds::total * (int)sizeof(double) : s->lds_bytes;
    int per_cu = (160 * 1024) / ldsb; if (per_cu > (seg20 ? 4 : 8)) per_cu = seg20 ? 4 : 8; if (per_cu < 1) per_cu = 1;
    int grid = s->num_cu * per_cu; if (grid > B) grid = B;
    int* ticket = B > 8 * grid ? s->d_ticket : nullptr;
    if (ticket && hipMemsetAsync(ticket, 0, sizeof(int), st) != hipSuccess) return admpc_set_error(ADMPC_EHIP, "hipMemsetAsync failed");
    if (seg20) {
        if (!s->d_slot && hipMalloc((void**)&s->d_slot, (size_t)s->num_cu * 4 * 2 * QSegLds::SLOT * sizeof(double)) != hipSuccess)
            return admpc_set_error(ADMPC_EHIP, "quad: slot buffer allocation failed");
        hipLaunchKernelGGL(admpc_quad_seg_kernel, dim3(grid), dim3(128), ldsb, st, s->d_cfg, B, x0, yref, yref_e, xbar, ubar, cost, status, iters, ticket, gp_state, route, which, s->d_slot);
    }
    else if (s->cfg.N * QU > 64)          // horizons beyond 16: one thread per input, two waves per instance
        hipLaunchKernelGGL(admpc_quad_solve_wide_kernel, dim3(grid), dim3(QW_NT), s->lds_bytes, st, s->d_cfg, B, x0, yref, yref_e, xbar, ubar, cost, status, iters, ticket, gp_state, route, which);
    else if (s->cfg.N * QU == 40 && !s->generic)
        hipLaunchKernelGGL(admpc_quad_solve_kernel<true>, dim3(grid), dim3(64), s->lds_bytes, st, s->d_cfg, B, x0, yref, yref_e, xbar, ubar, cost, status, iters, ticket, gp_state, route, which);
    else
        hipLaunchKernelGGL(admpc_quad_solve_kernel<false>, dim3(grid), dim3(64), s->lds_bytes, st, s->d_cfg, B, x0, yref, yref_e, xbar, ubar, cost, status, iters, ticket, gp_state, route, which);
    if (hipGetLastError() != hipSuccess) return admpc_set_error(ADMPC_EHIP, "quad solve kernel launch failed");
    return ADMPC_OK;
}

int admpc_quad_solve_batch_ex(AdmpcQuadSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* gp_state,
                              double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream)
{
    if (!s) return admpc_set_error(ADMPC_EINVAL, "null solver");
    if (B < 0) return admpc_set_error(ADMPC_EINVAL, "negative batch");
    if (B == 0) return ADMPC_OK;
    if (!x0 || !yref || !yref_e || !xbar || !ubar) return admpc_set_error(ADMPC_EINVAL, "null array argument");
    QGuard guard(s->device);
    if (!guard.good) return admpc_set_error(ADMPC_EHIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const int nsqp = s->cfg.sqp_iters > 1 ? s->cfg.sqp_iters : 1;
    if (nsqp == 1) return quad_solve(s, B, x0, yref, yref_e, gp_state, xbar, ubar, cost, status, iters, nullptr, 0, st);
    // ---- solver_type "SQP": nsqp QPs, acados' stopping test in front of every one but the first when a tolerance is set
    const int N = s->cfg.N;
    const bool tol_on = s->cfg.sqp_tol > 0.0;
    if (B > s->cap_sqp) {
        if (hipDeviceSynchronize() != hipSuccess) return admpc_set_error(ADMPC_EHIP, "hipDeviceSynchronize failed");
        for (int i = 0; i < 2; ++i) { if (s->d_A[i]) (void)hipFree(s->d_A[i]); if (s->d_B[i]) (void)hipFree(s->d_B[i]); s->d_A[i] = s->d_B[i] = nullptr; }
        if (s->d_phi) (void)hipFree(s->d_phi); if (s->d_act) (void)hipFree(s->d_act); if (s->d_fst) (void)hipFree(s->d_fst); if (s->d_st) (void)hipFree(s->d_st);
        s->d_phi = nullptr; s->d_act = s->d_fst = s->d_st = nullptr; s->cap_sqp = 0;
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i)
            ok = hipMalloc((void**)&s->d_A[i], (size_t)B * N * QX * QX * sizeof(double)) == hipSuccess && hipMalloc((void**)&s->d_B[i], (size_t)B * N * QX * QU * sizeof(double)) == hipSuccess;
        ok = ok && hipMalloc((void**)&s->d_phi, (size_t)B * N * QX * sizeof(double)) == hipSuccess && hipMalloc((void**)&s->d_act, (size_t)B * sizeof(int32_t)) == hipSuccess
                && hipMalloc((void**)&s->d_fst, (size_t)B * sizeof(int32_t)) == hipSuccess && hipMalloc((void**)&s->d_st, (size_t)B * sizeof(int32_t)) == hipSuccess;
        if (!ok) return admpc_set_error(ADMPC_EHIP, "quad SQP workspace allocation failed");
        s->cap_sqp = B;
    }
    int32_t* stat = status ? status : s->d_st;
    if (hipMemsetAsync(s->d_act, 0, (size_t)B * sizeof(int32_t), st) != hipSuccess || hipMemsetAsync(stat, 0, (size_t)B * sizeof(int32_t), st) != hipSuccess)
        return admpc_set_error(ADMPC_EHIP, "hipMemsetAsync failed");
    const double* gps = gp_state ? gp_state : x0;                    // the first node's GP state as the solve kernel takes it
    int gridS = s->num_cu * 2; if (gridS > B) gridS = B;
    const int tb = 64, tg = (B + tb - 1) / tb;
    for (int sq = 0; sq < nsqp; ++sq) {
        const int cur = sq & 1, prev = cur ^ 1;
        hipLaunchKernelGGL(admpc_quad_shoot_kernel, dim3(gridS), dim3(64), s->lds_bytes, st, s->d_cfg, B, (const double*)xbar, (const double*)ubar, gps, s->d_phi, s->d_A[cur], s->d_B[cur]);
        if (tol_on && sq > 0)
            hipLaunchKernelGGL(admpc_quad_sqp_test_kernel, dim3(tg), dim3(tb), 0, st, s->d_cfg, B, x0, yref, yref_e, (const double*)xbar, (const double*)ubar,
                               (const double*)s->d_A[prev], (const double*)s->d_B[prev], (const double*)s->d_phi, (const double*)s->d_A[cur], (const double*)s->d_B[cur], s->d_act, s->d_fst);
        int rc = quad_solve(s, B, x0, yref, yref_e, gp_state, xbar, ubar, cost, stat, iters, s->d_act, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(admpc_quad_sqp_merge_kernel, dim3(tg), dim3(tb), 0, st, B, (const int32_t*)stat, s->d_act, s->d_fst);
    }
    hipLaunchKernelGGL(admpc_quad_sqp_final_kernel, dim3(tg), dim3(tb), 0, st, B, tol_on ? 1 : 0, (const int32_t*)s->d_act, (const int32_t*)s->d_fst, stat);
    if (hipGetLastError() != hipSuccess) return admpc_set_error(ADMPC_EHIP, "quad SQP kernel launch failed");
    return ADMPC_OK;
}

int admpc_quad_solve_batch(AdmpcQuadSolver* s, int B, const double* x0, const double* yref, const double* yref_e,
                           double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream)
{
    return admpc_quad_solve_batch_ex(s, B, x0, yref, yref_e, nullptr, xbar, ubar, cost, status, iters, stream);
}

int admpc_quad_select_cluster_batch(int device, int B, int n_feat, const int32_t* feats, const double* x_sel, const double* u_sel,
                                    int K, const double* centroids, int32_t* route, void* stream)
{
    if (B < 0 || n_feat < 1 || n_feat > ADMPC_GP_MAX_FEAT || !feats || K < 1) return admpc_set_error(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!x_sel || !u_sel || !centroids || !route) return admpc_set_error(ADMPC_EINVAL, "null array argument");
    for (int j = 0; j < n_feat; ++j) if (feats[j] < 0 || feats[j] >= QY) return admpc_set_error(ADMPC_EINVAL, "feature index outside z = [x; u]");
    QGuard guard(device);
    if (!guard.good) return admpc_set_error(ADMPC_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(admpc_quad_select_cluster_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, n_feat, feats[0], n_feat > 1 ? feats[1] : 0,
                       n_feat > 2 ? feats[2] : 0, x_sel, u_sel, K, centroids, route);
    if (hipGetLastError() != hipSuccess) return admpc_set_error(ADMPC_EHIP, "quad select kernel launch failed");
    return ADMPC_OK;
}

int admpc_quad_solve_batch_routed(AdmpcQuadSolver* const* solvers, int K, int B, const int32_t* route,
                                  const double* x0, const double* yref, const double* yref_e, const double* gp_state,
                                  double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream)
{
    if (!solvers || K < 1) return admpc_set_error(ADMPC_EINVAL, "bad argument");
    if (B < 0) return admpc_set_error(ADMPC_EINVAL, "negative batch");
    if (B == 0) return ADMPC_OK;
    if (!route || !x0 || !yref || !yref_e || !xbar || !ubar) return admpc_set_error(ADMPC_EINVAL, "null array argument");
    for (int c = 0; c < K; ++c) {
        if (!solvers[c]) return admpc_set_error(ADMPC_EINVAL, "null solver");
        if (solvers[c]->device != solvers[0]->device || solvers[c]->cfg.N != solvers[0]->cfg.N) return admpc_set_error(ADMPC_EINVAL, "the cluster solvers must share device and horizon");
        if (solvers[c]->cfg.sqp_iters > 1) return admpc_set_error(ADMPC_EINVAL, "quad: solver_type SQP (sqp_iters > 1) is not implemented for routed (clustered GP) solves");
    }
    QGuard guard(solvers[0]->device);
    if (!guard.good) return admpc_set_error(ADMPC_EHIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(admpc_quad_route_invalid_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, route, K, status, cost, iters);
    for (int c = 0; c < K; ++c) {
        const int rc = quad_solve(solvers[c], B, x0, yref, yref_e, gp_state, xbar, ubar, cost, status, iters, route, c, st);
        if (rc) return rc;
    }
    return ADMPC_OK;
}

int admpc_quad_shoot_batch_ex(AdmpcQuadSolver* s, int B, const double* xbar, const double* ubar, const double* gp_state,
                              double* phi, double* A, double* Bm, void* stream)
{
    if (!s) return admpc_set_error(ADMPC_EINVAL, "null solver");
    if (B <= 0) return B == 0 ? ADMPC_OK : admpc_set_error(ADMPC_EINVAL, "negative batch");
    if (!xbar || !ubar || !phi || !A || !Bm) return admpc_set_error(ADMPC_EINVAL, "null array argument");
    QGuard guard(s->device);
    if (!guard.good) return admpc_set_error(ADMPC_EHIP, "hipSetDevice failed");
    int grid = s->num_cu * 2; if (grid > B) grid = B;
    hipLaunchKernelGGL(admpc_quad_shoot_kernel, dim3(grid), dim3(64), s->lds_bytes, (hipStream_t)stream, s->d_cfg, B, xbar, ubar, gp_state, phi, A, Bm);
    if (hipGetLastError() != hipSuccess) return admpc_set_error(ADMPC_EHIP, "quad shoot kernel launch failed");
    return ADMPC_OK;
}

int admpc_quad_shoot_batch(AdmpcQuadSolver* s, int B, const double* xbar, const double* ubar, double* phi, double* A, double* Bm, void* stream)
{
    return admpc_quad_shoot_batch_ex(s, B, xbar, ubar, nullptr, phi, A, Bm, stream);
}

#ifdef ADMPC_QUAD_TIMERS
#include <cstdio>
void admpc_quad_dump_timers(void)
{
    unsigned long long h[16] = {0};
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_quad_ticks), sizeof h) != hipSuccess) return;
    static const char* nm[11] = {"shooting", "condensing: setup + final store", "residuals + Newton matrix", "factorisation", "solves + step", "expansion", "-", "-",
                                 "condensing: propagation", "condensing: publish Gamma", "condensing: gradient + Hessian"};
    unsigned long long tot = 0; for (int i = 0; i < 11; ++i) tot += h[i];
    for (int i = 0; i < 11; ++i) if (h[i]) fprintf(stderr, "[quad phase] %-32s %14llu ticks %5.1f %%\n", nm[i], h[i], 100.0 * (double)h[i] / (double)(tot ? tot : 1));
}
#endif

}  // extern "C"
