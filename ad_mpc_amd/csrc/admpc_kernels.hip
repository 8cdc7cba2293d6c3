// admpc_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the batched AD-MPC solve engine, main translation unit.
//
// One SQP step:
//   A  admpc_linearize_kernel  one thread per (instance, stage, sensitivity-column group): ERK4 + forward
//                              sensitivities, writes the packed stage linearisation (42+7 values per stage)
//   then the QP of the step (H2-H6), one of
//   R  admpc_rowqp_kernel      (admpc_rowqp.hip) stage-wise Riccati interior point, one instance per 16-lane DPP row:
//                              every horizon, fp64 and fp32
//   C, D, E                    condensed pipeline for N = 20 fp64 (the reference's own QP strategy): condensing, dense
//                              LDL' interior point, expansion
//
// Hot path restated (reference = data_driven_mpc/ros_gp_mpc/src/ad_mpc/...):
//   H0/H1  model + ERK4 with forward sensitivities   ad_3d_optimizer.py:280-310, acados ERK
//                                                     (acados_solver_sim_car.c:655-665)
//   H2/H3  Gauss-Newton LS cost, soft/hard bounds     ad_3d_optimizer.py:146-199
//   H4/H5  QP solve: Mehrotra predictor-corrector primal-dual IPM (reference: full condensing + HPIPM,
//          acados_solver_sim_car.c:145,688-692; same unique minimiser)
//   H6     full step update of the iterate            acados_solver_sim_car.c:647-648,677
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <math.h>
#include "../../include/admpc.h"
#include "argmin_rule.h"

#define NX ADMPC_NX
#define NU ADMPC_NU
#define NY ADMPC_NY
#define WAVE 64
#define IPM_FLOOR 1e-40

namespace {

#include "model_dev.h"

// ---------------------------------------------------------------------------------------------
// kernel A: shooting + linearisation, one thread per (instance, stage, column group), T = double or float
// (storage and arithmetic).  Output, instance-major and packed as the QP kernels read it:
//   GTg [B][N][7][6]  stored columns c=0..6 <-> (A[:,2..6], B[:,0..1]), rows 0..5 of each column.
//                     Not stored because they are structural for this model (delta' = u1, positions
//                     do not feed back): A[:,0]=e0, A[:,1]=e1, row 6 of [A B] = [e6, 0, h].
//   blg [B][N][7]     defect b_k = phi(xbar_k,ubar_k) - xbar_{k+1}
// ---------------------------------------------------------------------------------------------
#define GTS 42           // values per stage of the packed linearisation
#define LIN_BLOCK 64     // threads per block of the linearisation kernel: single waves balance best over the CUs (256 registers each)
#define LIN_TASKS 63     // tasks per block (multiple of 3)
// work scheduler of the condensed path (int array): [0] ticket counter, [64 + q] number of instances in effort bucket q,
// [SCHED_HDR + q * cap + j] j-th instance of bucket q.  Zeroed by the linearisation kernel, filled by the condensing kernel,
// drained (highest bucket first) by the persistent interior-point waves.  Kernel R uses [0] as its ticket counter.
#define SCHED_NB 64
#define SCHED_HDR 128

template <class T>
__global__ __launch_bounds__(LIN_BLOCK) void admpc_linearize_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                              const T* __restrict__ xbarg, const T* __restrict__ ubarg,
                                                              const T* __restrict__ pg, const int32_t* __restrict__ skip,
                                                              T* __restrict__ GTg, T* __restrict__ blg, int* __restrict__ sched)
{
    const int N = cfg->N;
    const long total = (long)B * N * 3;
    if (sched && blockIdx.x == 0) for (int i = threadIdx.x; i < SCHED_HDR; i += blockDim.x) sched[i] = 0;     // ticket counter + bucket counts of this step
    // 63 tasks per 64-lane block: the three threads of a stage sit in adjacent lanes of one wave (gp_eval shares work among them)
    if (threadIdx.x >= LIN_TASKS) return;
    for (long tsk = (long)blockIdx.x * LIN_TASKS + threadIdx.x; tsk < total; tsk += (long)gridDim.x * LIN_TASKS) {
        const long sk = tsk / 3; const int g = (int)(tsk % 3);
        const long inst = sk / N; const int k = (int)(sk % N);
        if (skip && skip[inst] != 0) continue;
        T x[NX], u[NU], phi[NX], col[3][NX];
        const T* xs = xbarg + (inst * (N + 1) + k) * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xs[i];
        u[0] = ubarg[(inst * N + k) * NU]; u[1] = ubarg[(inst * N + k) * NU + 1];
        rk4_group<T>(cfg, x, u, pg[inst], (T)cfg->Ts, g, phi, col);
        T* GT = GTg + sk * GTS;
        const int c0 = g == 0 ? 0 : (g == 1 ? 3 : 5);
        const int nc = g == 0 ? 3 : 2;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            if (cc < nc)
#pragma unroll
                for (int i = 0; i < 6; ++i) GT[(c0 + cc) * 6 + i] = col[cc][i];
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) blg[sk * NX + i] = phi[i] - xs[NX + i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wave-level primitives
// ---------------------------------------------------------------------------------------------
#include "dense40.h"      // rdlane, WSYNC, lds_byte_addr, the 40 x 40 factorisation / substitution helpers, rcp_nr
#include "cond_common.h"  // div7, lane scans, stage_in / stage_dq, DenseLds


// ---------------------------------------------------------------------------------------------
// kernel B' : condensed QP, dense Cholesky -- the reference's own QP strategy (FULL_CONDENSING_HPIPM,
// acados_solver_sim_car.c:145) for horizons with 2N <= 64 inputs.  One instance per wavefront,
// lane i <-> input i = 2k+j: it owns row i of the condensed Hessian / of its Cholesky factor (in
// registers, statically indexed -> horizon is a template parameter) and the four inequalities of
// that input.  The states are eliminated: dx_k = xhat_k + Gamma_k du, so the interior-point state
// is (du, t, lam) only -- no dynamics multipliers, no state residuals.
//   H   = sum_k Gamma_k' Q_k Gamma_k               (input weights R are added on the diagonal on the fly)
//   g0  = r + sum_k Gamma_k' Q_k (xhat_k + xbar_k - xref_k)
//   delta row of stage k:  dx6_k = xhat_k[6] + h * sum_{k'<k} du_{(k',1)}   (structural: delta' = u1)
// ---------------------------------------------------------------------------------------------
// next instance for a persistent wave (wave-uniform), -1 when the step is drained.  Instances were binned by predicted
// interior-point effort; tickets walk the bins from the most expensive down (longest-processing-time-first), so the
// stragglers start early instead of last.  Lane l looks at bin SCHED_NB-1-l.
// The first ticket of a wave is its block index (no atomic: 2048 simultaneous draws on one word queue up for ~20 us),
// later ones are gridDim.x + a global counter.
// Placement (observed, scripts/probes/place_probe.hip + scripts/trace_d.py; nothing depends on it): the grid fills one wave
// per SIMD first, block b + 4*CUs lands on the SIMD of block b, and the wave that arrived first keeps full speed
// (11.1-11.9 us per iteration) while the second one gets 14-18 us as long as both are busy.  With first ticket = block index
// the 1024 predicted-longest instances are exactly the ones that run at full speed.
__device__ __forceinline__ int sched_next(int* __restrict__ sched, int cap, bool first) {
    const int lane = threadIdx.x;
    int t = blockIdx.x;
    if (!first) {
        int v = 0;
        if (lane == 0) v = atomicAdd(sched, 1);
        t = (int)gridDim.x + __builtin_amdgcn_readfirstlane(v);
    }
    const int c = sched[64 + SCHED_NB - 1 - lane];
    const int incl = wave_scan_incl_int(c);
    const unsigned long long m = __ballot(incl > t);
    if (m == 0ull) return -1;
    const int l = __ffsll((long long)m) - 1;
    const int base = __builtin_amdgcn_readlane(incl - c, l);
    return sched[SCHED_HDR + (size_t)(SCHED_NB - 1 - l) * cap + (t - base)];
}

// ---- optional in-kernel phase timers of the interior-point kernel (build with -DADMPC_PHASE_TIMERS; totals are printed by
//      admpc_destroy).  s_memtime ticks, summed over all waves: 0 staging, 1 phase A, 2 factorisation, 3 phase C,
//      4 substitutions, 5 expand/step, 6 final roll-out + outputs, 7 scheduler draw
#ifdef ADMPC_PHASE_TIMERS
__device__ unsigned long long g_phase_ticks[16];
__device__ __forceinline__ unsigned long long phase_now() {
    unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t;
}
#define PHASE_DECL() unsigned long long ph_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_last = phase_now()
#define PHASE_STAMP(k) do { const unsigned long long t_ = phase_now(); ph_acc[k] += t_ - ph_last; ph_last = t_; } while (0)
#define PHASE_FLUSH() do { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 16; ++q_) atomicAdd(&g_phase_ticks[q_], ph_acc[q_]); } } while (0)
#else
#define PHASE_DECL() do { } while (0)
#define PHASE_STAMP(k) do { } while (0)
#define PHASE_FLUSH() do { } while (0)
#endif

// ---- The four-kernel N = 20 pipeline of rounds 1-2 (kernels C, D, E behind kernel A): superseded by the fused persistent kernel
// (admpc_fused20.hip, whose phase bodies are these texts) and kept OUT of the product library -- `make legacy` builds
// ../libadmpc_legacy.so with -DADMPC_LEGACY_N20 for A/B runs (ADMPC_N20=split selects the pipeline there).
#ifdef ADMPC_LEGACY_N20
// kernel C (N = 20 path): condensing.  One instance per wavefront, lane i <-> input i.  Writes, per instance, the packed
// lower-triangular Hessian rows H[NTRI] and aux[128] = { g0[64] (reduced gradient at du = 0, per input), xhat6[64] (free
// response of delta per stage) } for the interior-point kernel.  A kernel of its own so that its 40-double Hessian row and
// the IPM state never compete for registers (and so that the IPM kernel's code stays small).
// QMASK: state components that may carry a tracking weight (bit c <-> W[c] or We[c] non-zero); the host picks 0b0000111
// (position + heading, the reference's weights) or the general 0b1111111 instantiation, which tests the weights at run time.
template <int NT, int QMASK>
__global__ __launch_bounds__(WAVE, 2) void admpc_condense_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                              const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                              const double* __restrict__ yrefeg,
                                                              const double* __restrict__ GTg, const double* __restrict__ blg,
                                                              const double* __restrict__ xbarg, const double* __restrict__ ubarg,
                                                              const int32_t* __restrict__ statusg, int first_pass,
                                                              double* __restrict__ Hg, double* __restrict__ auxg,
                                                              int* __restrict__ sched, int cap)
{
    constexpr int N = NT, n = 2 * NT, NTRI = DenseLds<NT>::NTRI;
    extern __shared__ double lds_raw[];
    double* const Hp = lds_raw;
    double* const GT = Hp + NTRI + (NTRI & 1);
    double* const bl = GT + N * GTS;
    double* const dq = bl + DenseLds<NT>::BLS;
    double* const gam = dq + DenseLds<NT>::DQS;    // [NX][64] Gamma components of the current stage, lane = input
    const int lane = threadIdx.x;
    const int ki = lane >> 1, ji = lane & 1;
    const bool uact = lane < n;
    const double Ts = cfg->Ts, h = cfg->Ts;
    double Qd[NX], Qe[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { Qd[i] = Ts * cfg->W[i]; Qe[i] = cfg->We[i]; }
    const double Rj = Ts * cfg->W[NX + ji];
    PHASE_DECL();
    for (int inst = blockIdx.x; inst < B; inst += gridDim.x) {
        if (!first_pass && statusg[inst] != 0) continue;
        const double* xbg = xbarg + (size_t)inst * (N + 1) * NX;
        const double* ubg = ubarg + (size_t)inst * N * NU;
        const double* yrg = yrefg + (size_t)inst * N * NY;
        const double* gtg = GTg + (size_t)inst * N * GTS;
        PHASE_STAMP(13);
        // ---------------- stage the instance ----------------
        stage_in<N * GTS>(GT, gtg, lane);
        stage_in<N * NX>(bl, blg + (size_t)inst * N * NX, lane);
        stage_dq<N>(dq, xbg, yrg, yrefeg + (size_t)inst * NX, lane);
        const int sc = uact ? lane : 0;
        const double ubar_i = ubg[sc];
        const double r_i = Rj * (ubar_i - yrg[(sc >> 1) * 9 + 7 + (sc & 1)]);
        double xh[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) xh[c] = x0g[(size_t)inst * NX + c] - xbg[c];      // uniform
        WSYNC();
        PHASE_STAMP(10);
        // ---------------- condensing ----------------
        double g[NX], hrow[n];
#pragma unroll
        for (int c = 0; c < NX; ++c) g[c] = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) hrow[i] = 0.0;
        double g0 = r_i;
        double xh6_own = 0.0;                   // xhat_k[6] of the stage whose delta box this lane owns
        static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            constexpr int lim = 2 * k < n ? 2 * k : n;        // inputs of stages < k (even)
            constexpr int nblk = (lim + 15) / 16;             // 16-lane blocks of Gamma that are non-zero at this stage
            // One stage = one basic block: merged into a single 21-stage block, hipcc hoists every LDS load of the whole instance
            // and spills ~1600 registers.  The test is always true (B >= 1) but opaque to the compiler.
            int tok = B; asm volatile("" : "+s"(tok));
            if (tok > 0) {
            double wg[NX], Rb[NX][3];
            if constexpr (k >= 1) {
                // ---- cost of stage k, part 1: g0 += Gamma_k' Q (xhat + dq); publish the weighted components of this lane's Gamma
                //      column in LDS and start reading them back as "block m of component c in every 16-lane row" (the DPP
                //      sources of part 2).  The reads complete under the propagation below.
                if (lane == k) xh6_own = xh[6];
                static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr ((QMASK >> c) & 1) {
                        const double w = k < N ? Qd[c] : Qe[c];
                        wg[c] = w * g[c];
                        g0 += wg[c] * (xh[c] + dq[k * 7 + c]);
                        gam[c * 64 + lane] = g[c];
                    }
                });
                static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr ((QMASK >> c) & 1) {
#pragma unroll
                        for (int m = 0; m < nblk; ++m) Rb[c][m] = gam[c * 64 + 16 * m + (lane & 15)];
                    }
                });
            }
            PHASE_STAMP(11);
            double xn[NX], gn[NX];
            if constexpr (k < N) {
                // ---- propagate: xhat_{k+1} = A xhat + b ; Gamma_{k+1}[:,i] = A Gamma_k[:,i]  (or B[:,j] for the inputs of stage k)
                const double* Gk = GT + k * GTS;
#pragma unroll
                for (int r = 0; r < 6; ++r) { xn[r] = bl[k * 7 + r] + (r < 2 ? xh[r] : 0.0); gn[r] = r < 2 ? g[r] : 0.0; }
                xn[6] = bl[k * 7 + 6] + xh[6]; gn[6] = g[6];
#pragma unroll
                for (int c = 0; c < 5; ++c) {
#pragma unroll
                    for (int r = 0; r < 6; r += 2) {
                        const double2 a = *reinterpret_cast<const double2*>(Gk + c * 6 + r);
                        xn[r] += a.x * xh[c + 2]; xn[r + 1] += a.y * xh[c + 2];
                        gn[r] += a.x * g[c + 2];  gn[r + 1] += a.y * g[c + 2];
                    }
                }
                const bool mine = ki == k;
                // B_k columns: loaded by every lane and pinned by an empty asm -- left alone, hipcc sinks each load into a divergent
                // "if (mine)" block of its own (branch + ds_read + full wait, six times per stage)
                double bb[12];
#pragma unroll
                for (int r = 0; r < 12; r += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(Gk + 5 * 6 + r);
                    bb[r] = v.x; bb[r + 1] = v.y;
                }
#pragma unroll
                for (int r = 0; r < 12; ++r) asm volatile("" : "+v"(bb[r]));
#pragma unroll
                for (int r = 0; r < 6; ++r) gn[r] = mine ? (ji ? bb[6 + r] : bb[r]) : gn[r];
                gn[6] = mine ? (ji ? h : 0.0) : gn[6];
            }
            PHASE_STAMP(12);
            if constexpr (k >= 1) {
                // ---- cost of stage k, part 2: H += Gamma_k' Q Gamma_k over the inputs of stages < k
                static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr ((QMASK >> c) & 1) {
                        static_for<0, lim / 4>([&](auto q) __attribute__((always_inline)) {
                            constexpr int i2 = 4 * decltype(q)::value;
                            fmac_rowbc4_ld<i2 % 16>(hrow[i2], hrow[i2 + 1], hrow[i2 + 2], hrow[i2 + 3], Rb[c][i2 / 16], wg[c]);
                        });
                        if constexpr (lim % 4 == 2) {
                            fmac_rowbc_ld<(lim - 2) % 16>(hrow[lim - 2], Rb[c][(lim - 2) / 16], wg[c]);
                            fmac_rowbc_ld<(lim - 1) % 16>(hrow[lim - 1], Rb[c][(lim - 1) / 16], wg[c]);
                        }
                    }
                });
            }
            if constexpr (k < N) {
#pragma unroll
                for (int r = 0; r < NX; ++r) { g[r] = gn[r]; xh[r] = xn[r]; }
            }
            }
            PHASE_STAMP(13);
        });
        // packed lower-triangular rows of H into LDS
        static_assert(n == 40, "row store assembly is generated for n = 40");
        store_row_40(hrow, lds_byte_addr(Hp + (uact ? (lane * (lane + 1)) / 2 : 0)));
        WSYNC();
        stage_in<NTRI>(Hg + (size_t)inst * NTRI, Hp, lane);
        auxg[(size_t)inst * 128 + lane] = g0;
        auxg[(size_t)inst * 128 + 64 + lane] = xh6_own;
        // ---- effort bin for the scheduler: how far the diagonally scaled gradient step -g0_i / (H_ii + R_i) overshoots the input
        //      box, relative to the box width (max over inputs).  0 = no bound in sight (4-5 interior-point iterations); the
        //      iteration count grows with it (correlation 0.86 on the config-2 scenarios).  A heuristic: it orders work, nothing else.
        {
            const double hii = Hp[uact ? (lane * (lane + 1)) / 2 + lane : 0] + Rj;
            const double sstep = -g0 / hii;
            const double over = fmax(sstep - (cfg->ubu[ji] - ubar_i), (cfg->lbu[ji] - ubar_i) - sstep) / (cfg->ubu[ji] - cfg->lbu[ji]);
            const double score = wave_reduce<OpMax>(uact ? fmax(over, 0.0) : 0.0);
            const int q = (int)fmin(fmax(score * 32.0, 0.0), (double)(SCHED_NB - 1));
            if (lane == 0) {
                const int pos = atomicAdd(sched + 64 + q, 1);
                sched[SCHED_HDR + (size_t)q * cap + pos] = inst;
            }
        }
        WSYNC();
    }
    PHASE_FLUSH();
}

template <int NT>
__global__ __launch_bounds__(WAVE, 2) void admpc_qp_dense_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                                 const double* __restrict__ xbarg, const double* __restrict__ ubarg,
                                                                 double* __restrict__ costg, int32_t* __restrict__ statusg,
                                                                 int32_t* __restrict__ itersg,
                                                                 const double* __restrict__ Hg, double* auxg_out,
                                                                 int* __restrict__ sched, int cap)
{
    const double* auxg = auxg_out;              // in: g0[64] | xhat6[64] per instance; out: du[64] over the g0 slot
    constexpr int N = NT, n = 2 * NT, NTRI = DenseLds<NT>::NTRI;
    extern __shared__ double lds_raw[];
    double* const Hp = lds_raw;                 // packed lower-triangular rows of H
    double* const Lp = Hp + NTRI + (NTRI & 1);  // packed strictly-lower rows of the unit factor L (M = L D L')
    double* const park = Lp + NTRI + (NTRI & 1);   // [5][64] per-lane constants (registers are the scarce resource)
    double* const cb = park + 5 * 64;           // [64] step broadcast buffer
    double* const invd = cb + 64;               // [64] 1 / D_jj
    double* const sb = invd + 64;               // [64] per-stage exchange
    double* const sb2 = sb + 64;                // [64]
#define PK_DL   park[0 * 64 + lane]
#define PK_DUU  park[1 * 64 + lane]
#define PK_G0   park[2 * 64 + lane]
#define PK_DDL  park[3 * 64 + lane]
#define PK_DDU  park[4 * 64 + lane]

    const int lane = threadIdx.x;
    const int ki = lane >> 1, ji = lane & 1;
    const bool uact = lane < n;
    const bool dact = lane >= 1 && lane < N;
    const double Ts = cfg->Ts, h = cfg->Ts;
    const double Rj = Ts * cfg->W[NX + ji];
    const double rho_l = Ts * cfg->zl, rho_u = Ts * cfg->zu;
    const double thr = cfg->ipm_thr0, mu0 = cfg->ipm_mu0;
    const double tol_comp = cfg->ipm_tol_comp, tol_res = cfg->ipm_tol_res, tol_step = cfg->ipm_tol_step;
    const int itmax = cfg->ipm_iter_max;
    const bool try_unc = cfg->ipm_try_unconstrained != 0.0;
    const double thw = cfg->ipm_warm_thr, wrest = cfg->ipm_warm_restart;
    const int fbit = (int)cfg->ipm_fallback_iter;
    const double inv_nineq = 1.0 / (double)(8 * N + 2 * (N - 1));
    // Factorisation of the Newton matrix M = H + diag(dbar) + (s_odd on the odd columns of the u1 rows) into L D L' (LDS: Lp, invd).
    // Used twice per instance at most: once without barrier terms (the unconstrained trial) and once per interior-point iteration.
    auto factorise = [&](const double dbar_, const double sodd_, const int lz_) __attribute__((always_inline)) {
        const int trz_ = lz_ * (lz_ + 1) / 2;
        const bool uz_ = lz_ < n;
        double a[n];
        // a[c] = (c <= lane ? H[lane][c] : 0) + (c odd and this lane is a u1 input ? S_i : 0) + (c == lane ? Dbar : 0);
        // idle lanes: unit rows (Dbar = 1 on a diagonal that never becomes a pivot, zeros elsewhere)
        newton_row_40(a, lds_byte_addr(Hp + (uz_ ? trz_ : 0)), dbar_, sodd_);
        // Square-root-free right-looking factorisation M = L D L' (unit lower L), row i in the registers of lane i.
        // Column j (unscaled, w_i = a_i[j]) is published in LDS and read back as "block m in every 16-lane row", the DPP
        // sources of the rank-1 update: a[jj] -= w_jj * (w_i / D_jj) is ONE v_fmac_f64_dpp per jj.  Look-ahead: the first
        // update of column j makes column j+1 final; its pivot chain (v_readlane, v_rcp_f64 + Newton, scale, publish,
        // read back) is started right there and completes under the remaining updates of column j.
        // Entries on and above the diagonal of a row are never read (lane jj's w_jj is only picked up for jj > j), so the
        // column is used unmasked; only the store of L is masked (EXEC).
        const unsigned lrow = lds_byte_addr(Lp + (uz_ ? trz_ : 0));
        const unsigned pub_wr = lds_byte_addr(cb + lane), pub_rd = lds_byte_addr(cb + (lane & 15));
        // pivot chain of column j: reciprocal of the pivot, scaled column (the factor's entries), its masked store
        auto chain = [&](auto jc, double& nln) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const double dj = rdlane(a[j], j);
            const double dinv = rcp_nr(dj);                             // 1 / D_jj
            const double lu = a[j] * dinv;                              // L_ij for the lanes below the diagonal
            invd[j] = dinv;                                             // uniform value, same address
            if constexpr (j + 1 < n) {
                asm volatile("s_bfm_b64 exec, %2, %3\n\tds_write_b64 %0, %1 offset:%4\n\ts_mov_b64 exec, -1"
                             : : "v"(lrow), "v"(lu), "n"(n - 1 - j), "n"(j + 1), "n"(8 * j) : "memory");
                nln = -lu;
            }
        };
        // column 0: published by plain code (nothing to overlap with yet).  The block registers ping-pong between two sets so that
        // no copy ever reads a register whose LDS load is still in flight.
        double Rb[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}, nlb[2] = {0.0, 0.0};
        cb[lane] = a[0];
#pragma unroll
        for (int m = 0; m < 3; ++m) Rb[0][m] = cb[16 * m + (lane & 15)];
        chain(std::integral_constant<int, 0>{}, nlb[0]);
        static_for<0, n - 1>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr bool own = (j + 1) / 16 == 2;                     // column j: DPP sources are the lanes' own registers (VALU-written)
            constexpr bool pub = j + 2 < n && (j + 2) / 16 < 2;         // column j + 1 still needs its blocks in other rows
            double (&R)[3] = Rb[j & 1];
            double (&Rn)[3] = Rb[(j + 1) & 1];
            double& nl = nlb[j & 1];
            double& nln = nlb[(j + 1) & 1];
            col_head<j + 1, (j + 2) / 16, pub, own>(a[j + 1], R, nl, Rn, pub_wr, pub_rd);
            if constexpr (!pub) Rn[2] = a[j + 1];                       // only lanes 32..39 are still involved: own row
            chain(std::integral_constant<int, j + 1>{}, nln);
            constexpr int j4 = ((j + 2 + 3) / 4) * 4 < n ? ((j + 2 + 3) / 4) * 4 : n;        // first 4-aligned column >= j + 2
            static_for<j + 2, j4>([&](auto c) __attribute__((always_inline)) {
                constexpr int jj = decltype(c)::value;
                if constexpr (own) fmac_rowbc<jj % 16>(a[jj], R[jj / 16], nl);
                else fmac_rowbc_ld<jj % 16>(a[jj], R[jj / 16], nl);
            });
            static_for<j4 / 4, n / 4>([&](auto c) __attribute__((always_inline)) {
                constexpr int jj = 4 * decltype(c)::value;
                if constexpr (own) fmac_rowbc4<jj % 16>(a[jj], a[jj + 1], a[jj + 2], a[jj + 3], R[jj / 16], nl);
                else fmac_rowbc4_ld<jj % 16>(a[jj], a[jj + 1], a[jj + 2], a[jj + 3], R[jj / 16], nl);
            });
        });
        WSYNC();
    };
    // M x = y through the factor: L z = y, z *= D^-1, L' x = z (assembly, see gen_subst_asm.py)
    auto ldl_solve = [&](double y, const int lz_) __attribute__((always_inline)) -> double {
        static_assert(n == 40, "the substitution assembly is generated for n = 40");
        const bool uz_ = lz_ < n;
        const unsigned pub = lds_byte_addr(cb + (lz_ & 15));                              // cb is free while a system is being solved
        fwd_subst_40(y, lds_byte_addr(Lp + (uz_ ? lz_ * (lz_ + 1) / 2 : 0)), pub);      // idle lanes never take part (EXEC masks)
        double x = y * invd[uz_ ? lz_ : 0];
        bwd_subst_40(x, lds_byte_addr(Lp + (uz_ ? lz_ : 0)), pub);
        return x;
    };

    // Instances need 4 .. 15+ interior-point iterations each: a static instance -> wave map leaves most of the chip idle while
    // the unlucky waves finish.  The waves draw instances from the scheduler, predicted-expensive ones first.
    // (Instances that failed in an earlier SQP iteration were not queued by the condensing kernel.)
    // diagonal slots of the packed factor: 0.0, never overwritten (the factorisation stores the strictly-lower part only).  The
    // substitution assembly lets the source lane of a step take part with this multiplier.
    if (uact) Lp[(lane * (lane + 1)) / 2 + lane] = 0.0;
    WSYNC();
    PHASE_DECL();
    for (int inst = sched_next(sched, cap, true); inst >= 0; inst = sched_next(sched, cap, false)) {
        PHASE_STAMP(7);
#ifdef ADMPC_TRACE_SCHED   // debug build (scripts/trace_d.py): start time, duration, block and iteration count of every instance, packed into `cost`
        unsigned long long dbg_t0; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dbg_t0));
#endif
        const double* xbg = xbarg + (size_t)inst * (N + 1) * NX;
        const double* ubg = ubarg + (size_t)inst * N * NU;
        // ---------------- stage: condensed Hessian (kernel C) and per-lane data ----------------
        stage_in<NTRI>(Hp, Hg + (size_t)inst * NTRI, lane);
        const int sc = uact ? lane : 0;
        const double ubar_i = ubg[sc];
        const double dl_i = cfg->lbu[ji] - ubar_i, duu_i = cfg->ubu[ji] - ubar_i;
        const double g0 = auxg[(size_t)inst * 128 + lane];
        const double xh6_own = auxg[(size_t)inst * 128 + 64 + lane];
        // ---------------- interior point start ----------------
        double t[4], lam[4], du = 0.0, sl = thr, su = thr;
        {
            const double r0[4] = { thr - dl_i, thr + duu_i, thr, thr };
#pragma unroll
            for (int i = 0; i < 4; ++i) { t[i] = r0[i] > thr ? r0[i] : thr; lam[i] = mu0 * rcp_nr(t[i]); }
        }
        double Dt[2] = {1.0, 1.0}, Dlam[2] = {0.0, 0.0}, Ddl = 0.0, Ddu = 0.0, dx6 = 0.0;
        if (dact) {
            const double x6 = xbg[lane * 7 + 6];
            Ddl = cfg->lbx_delta - x6; Ddu = cfg->ubx_delta - x6;
            dx6 = xh6_own;
            const double r0[2] = { dx6 - Ddl, Ddu - dx6 };
#pragma unroll
            for (int i = 0; i < 2; ++i) { Dt[i] = r0[i] > thr ? r0[i] : thr; Dlam[i] = mu0 * rcp_nr(Dt[i]); }
        }
        WSYNC();
        PK_DL = dl_i; PK_DUU = duu_i; PK_G0 = g0; PK_DDL = Ddl; PK_DDU = Ddu;      // parked in LDS: registers are the scarce resource
        WSYNC();

        bool failed = false;
        double rmax_prev = 0.0, step = 1e300, stp_local = 1e300, alpha_prev = 1.0;
        int it = 0;
        PHASE_STAMP(0);
        // ---------------- trial: the QP without its inequalities ----------------
        // (H + R) du = -g0 is one factorisation and one solve (about 0.6 of an interior-point iteration).  If that minimiser
        // respects the input box and the steering box it is the solution of the full QP -- no bound is active, the slacks are
        // zero -- and the interior point is skipped (iters = 0).  True for 55 % of the config-2 scenarios; the oracle does the same.
        bool solved = false, warmed = false, cons = false;
        if (try_unc) {
            int lt = lane; asm volatile("" : "+v"(lt));
            factorise(uact ? Rj : 1.0, 0.0, lt);
            const double xt = ldl_solve(uact ? -g0 : 0.0, lt);
            const double duc = uact ? xt : 0.0;
            cb[lane] = duc;
            WSYNC();
            const double du1_stage = lane < N ? cb[2 * lane + 1] : 0.0;
            const double pre = wave_scan_incl<OpSum>(du1_stage);
            const double dx6c = xh6_own + h * (pre - du1_stage);
            const bool ok = (!uact || (duc >= dl_i && duc <= duu_i)) && (!dact || (dx6c >= Ddl && dx6c <= Ddu));
            WSYNC();
            if (__all(ok)) { du = duc; solved = true; }
            else if (thw > 0.0) {
                // warm start (cfg.ipm_warm_thr): the interior point starts from that minimiser.  A violated input bound is absorbed
                // by its slack (the input box is soft), a violated steering bound stays as a primal residual.
                warmed = true;
                du = duc;
                sl = fmax(dl_i - duc, 0.0) + thw; su = fmax(duc - duu_i, 0.0) + thw;
                const double r0[4] = { duc + sl - dl_i, su + duu_i - duc, sl, su };
#pragma unroll
                for (int i = 0; i < 4; ++i) { t[i] = r0[i] > thw ? r0[i] : thw; lam[i] = mu0 * rcp_nr(t[i]); }
                if (dact) {
                    dx6 = dx6c;
                    const double q0[2] = { dx6 - Ddl, Ddu - dx6 };
#pragma unroll
                    for (int i = 0; i < 2; ++i) { Dt[i] = q0[i] > thw ? q0[i] : thw; Dlam[i] = mu0 * rcp_nr(Dt[i]); }
                }
            }
        }
        PHASE_STAMP(8);
        // the cold start of the interior point, in place (wave-uniform callers): cfg.ipm_warm_restart and cfg.ipm_fallback_iter
        auto cold_start = [&]() __attribute__((always_inline)) {
            const double dlc = PK_DL, duc2 = PK_DUU;
            du = 0.0; sl = thr; su = thr;
            const double r0[4] = { thr - dlc, thr + duc2, thr, thr };
#pragma unroll
            for (int i = 0; i < 4; ++i) { t[i] = r0[i] > thr ? r0[i] : thr; lam[i] = mu0 * rcp_nr(t[i]); }
            dx6 = dact ? xh6_own : 0.0;
            const double q0[2] = { dx6 - PK_DDL, PK_DDU - dx6 };
#pragma unroll
            for (int i = 0; i < 2; ++i) { Dt[i] = dact ? (q0[i] > thr ? q0[i] : thr) : 1.0; Dlam[i] = dact ? mu0 * rcp_nr(Dt[i]) : 0.0; }
            alpha_prev = 1.0; stp_local = 1e300;
        };
        if (!solved)
        for (; it < itmax + (cons ? fbit : 0); ++it) {
            int lz = lane;                          // laundered lane id: per-lane addresses / predicates derived from it are recomputed in
            asm volatile("" : "+v"(lz));            // place instead of being hoisted out of the loops (hipcc parked ~200 of them in scratch)
            const int trz = lz * (lz + 1) / 2;
            const bool uz = lz < n;
            // ---- phase A: complementarity, reduced gradient, convergence test, Newton matrix, Cholesky.
            // Everything derived from (t, lam) in this phase dies before the factorisation ends: the 40-double factor row
            // and the interior-point state must not be live at the same time (256-register budget, two waves per SIMD).
            double ru, mu, Dbar, S_i;
            {
                double musum = 0.0, cmax = 0.0, rmax = 0.0;
                double G0, G1, G2, G3;
                {
                    const double i0 = rcp_nr(t[0]), i1 = rcp_nr(t[1]), i2_ = rcp_nr(t[2]), i3 = rcp_nr(t[3]);
                    G0 = lam[0] * i0; G1 = lam[1] * i1; G2 = lam[2] * i2_; G3 = lam[3] * i3;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { const double rci = t[i] * lam[i]; musum += uact ? rci : 0.0; cmax = fmax(cmax, uact ? rci : 0.0); }
#pragma unroll
                for (int i = 0; i < 2; ++i) { const double rci = Dt[i] * Dlam[i]; musum += dact ? rci : 0.0; cmax = fmax(cmax, dact ? rci : 0.0); }
                const double G56 = Dlam[0] * rcp_nr(Dt[0]) + Dlam[1] * rcp_nr(Dt[1]);
                Dbar = uact ? Rj + G0 * G2 * rcp_nr(G0 + G2) + G1 * G3 * rcp_nr(G1 + G3) : 1.0;       // idle lanes: identity rows
                // reduced gradient  ru = H du + R du + g0 - lam0 + lam1 + [u1 inputs] h * sum_{k>ki} (lam6_k - lam5_k)
                cb[lane] = uact ? du : 0.0;
                const double dlam_pref = wave_scan_incl<OpSum>(dact ? (Dlam[1] - Dlam[0]) : 0.0);     // lanes = stages
                sb[lane] = rdlane(dlam_pref, 63) - dlam_pref;           // suffix over stages > lane
                const double Ssuf_incl = wave_scan_incl<OpSum>(dact ? G56 : 0.0);
                sb2[lane] = rdlane(Ssuf_incl, 63) - Ssuf_incl;          // lane = stage: sum over stages > lane
                WSYNC();
                // H du: the lane's full row of H by EXEC-masked loads (no address arithmetic), du[c] through DPP row broadcasts
                double hdu = 0.0;
                {
                    double Rd3[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) Rd3[m] = cb[16 * m + (lane & 15)];
                    double hv[n];
                    sym_row_40(hv, lds_byte_addr(Hp + (uz ? trz : 0)), lds_byte_addr(Hp + (uz ? lz : 0)));
                    static_for<0, n>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        fmac_rowbc_ld<c % 16>(hdu, Rd3[c / 16], hv[c]);
                    });
                }
                ru = hdu + Rj * du + PK_G0 - lam[0] + lam[1] + (ji ? h * sb[uact ? ki : 0] : 0.0);
                S_i = h * h * sb2[uact ? ki : 0];                        // lane = input: S_{k_i}
                {
                    const double rd0 = du + sl - PK_DL - t[0], rd1 = -du + su + PK_DUU - t[1], rd2 = sl - t[2], rd3 = su - t[3];
                    const double rsl = rho_l - lam[0] - lam[2], rsu = rho_u - lam[1] - lam[3];
                    const double Drd0 = dx6 - PK_DDL - Dt[0], Drd1 = PK_DDU - dx6 - Dt[1];
                    double ra = OpMaxNan::f(fabs(ru), fabs(rsl)); ra = OpMaxNan::f(ra, fabs(rsu));
                    ra = OpMaxNan::f(ra, fabs(rd0)); ra = OpMaxNan::f(ra, fabs(rd1)); ra = OpMaxNan::f(ra, fabs(rd2)); ra = OpMaxNan::f(ra, fabs(rd3));
                    const double rb = OpMaxNan::f(fabs(Drd0), fabs(Drd1));
                    rmax = OpMaxNan::f(uact ? ra : 0.0, dact ? rb : 0.0);
                }
                mu = wave_reduce<OpSum>(musum) * inv_nineq;
                cmax = wave_reduce<OpMax>(cmax);
                rmax = wave_reduce<OpMaxNan>(rmax);
                step = wave_reduce<OpMax>(stp_local);
                if (!(mu == mu) || !(rmax == rmax)) { failed = true; break; }
                if (cmax <= tol_comp && step <= tol_step &&
                        (rmax <= tol_res || (it > 0 && rmax > 0.1 * rmax_prev && rmax <= ADMPC_IPM_FLOOR_CAP * tol_res))) break;      // admpc.h: stopping test
                rmax_prev = rmax;
            }
            if (fbit > 0 && !cons && it >= fbit) {
                // cfg.ipm_fallback_iter: still iterating, most likely in a limit cycle of the centring heuristic.  Start over and finish
                // with plain predictor-centring steps (no second-order term) on a budget of its own; this pass is redone from the cold start.
                cons = true; warmed = false;
                cold_start();
                rmax_prev = 0.0;
                --it;
                continue;
            }
            PHASE_STAMP(1);
            // ---- Newton matrix row: M = H + diag(R + barrier) + h^2 S_{max(k,k')} on the u1 x u1 block, then its factorisation
            factorise(Dbar, (uz && ji) ? S_i : 0.0, lz);
            PHASE_STAMP(2);
            // ---- phase C: re-derive the barrier quantities from (t, lam).  The asm statements make the compiler forget what
            //      it computed in phase A so that nothing but the state itself stays live across the factorisation.
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(t[i]), "+v"(lam[i]));
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(Dt[i]), "+v"(Dlam[i]));
            double it_[4], il_[4], rc[4], Dit[2], Dil[2], Drc[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { it_[i] = rcp_nr(t[i]); il_[i] = rcp_nr(lam[i]); rc[i] = t[i] * lam[i]; }
#pragma unroll
            for (int i = 0; i < 2; ++i) { Dit[i] = rcp_nr(Dt[i]); Dil[i] = rcp_nr(Dlam[i]); Drc[i] = Dt[i] * Dlam[i]; }
            const double G0 = lam[0] * it_[0], G1 = lam[1] * it_[1], G2 = lam[2] * it_[2], G3 = lam[3] * it_[3];
            const double iG02 = rcp_nr(G0 + G2), iG13 = rcp_nr(G1 + G3);
            const double G5 = Dlam[0] * Dit[0], G6 = Dlam[1] * Dit[1];
            const double rd0 = du + sl - PK_DL - t[0], rd1 = -du + su + PK_DUU - t[1], rd2 = sl - t[2], rd3 = su - t[3];
            const double rsl = rho_l - lam[0] - lam[2], rsu = rho_u - lam[1] - lam[3];
            const double Drd0 = dx6 - PK_DDL - Dt[0], Drd1 = PK_DDU - dx6 - Dt[1];

            double mu_aff = 0.0, dsl = 0.0, dsu = 0.0, ddu = 0.0, dt[4], dlam[4], Ddt[2], Ddlam[2];
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                // ---- right-hand side: eliminate slacks / multipliers
                const double c0 = rc[0] * it_[0], c1 = rc[1] * it_[1], c2 = rc[2] * it_[2], c3 = rc[3] * it_[3];
                const double e1 = rsl + c0 + c2 + G0 * rd0 + G2 * rd2;
                const double e2 = rsu + c1 + c3 + G1 * rd1 + G3 * rd3;
                const double etal = c0 + G0 * rd0 - G0 * e1 * iG02;
                const double etau = -c1 - G1 * rd1 + G1 * e2 * iG13;
                const double ek = dact ? (Drc[0] * Dit[0] + G5 * Drd0) - (Drc[1] * Dit[1] + G6 * Drd1) : 0.0;
                const double epref = wave_scan_incl<OpSum>(ek);
                sb[lane] = rdlane(epref, 63) - epref;
                WSYNC();
                double y = uact ? -(ru + etal + etau + (ji ? h * sb[ki] : 0.0)) : 0.0;
                PHASE_STAMP(pass == 0 ? 3 : 5);
                // ---- L z = y, z *= D^-1, L' x = z  (unit lower L packed by rows in LDS; assembly, see gen_subst_asm.py)
                const double x = ldl_solve(y, lz);
                PHASE_STAMP(4);
                ddu = uact ? x : 0.0;
                // ---- delta rows: ddx6_k = h * sum_{k'<k} ddu_{(k',1)}
                cb[lane] = ddu;
                WSYNC();
                const double du1_stage = lane < N ? cb[2 * lane + 1] : 0.0;
                const double pre = wave_scan_incl<OpSum>(du1_stage);
                const double ddx6 = h * (pre - du1_stage);
                // ---- expand
                dsl = -(e1 + G0 * ddu) * iG02;
                dsu = -(e2 - G1 * ddu) * iG13;
                dt[0] = ddu + dsl + rd0; dt[1] = -ddu + dsu + rd1; dt[2] = dsl + rd2; dt[3] = dsu + rd3;
                const double Gs[4] = { G0, G1, G2, G3 };
                double rr = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dlam[i] = -rc[i] * it_[i] - Gs[i] * dt[i];
                    rr = fmax(rr, uact ? fmax(-dt[i] * it_[i], -dlam[i] * il_[i]) : 0.0);
                }
                Ddt[0] = ddx6 + Drd0;  Ddlam[0] = -Drc[0] * Dit[0] - G5 * Ddt[0];
                Ddt[1] = -ddx6 + Drd1; Ddlam[1] = -Drc[1] * Dit[1] - G6 * Ddt[1];
#pragma unroll
                for (int i = 0; i < 2; ++i) rr = fmax(rr, dact ? fmax(-Ddt[i] * Dit[i], -Ddlam[i] * Dil[i]) : 0.0);
                rr = wave_reduce<OpMax>(rr);
                const double amax = rr > 1.0 ? rcp_nr(rr) : 1.0;
                if (pass == 0) {
                    double s_aff = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) s_aff += uact ? (t[i] + amax * dt[i]) * (lam[i] + amax * dlam[i]) : 0.0;
#pragma unroll
                    for (int i = 0; i < 2; ++i) s_aff += dact ? (Dt[i] + amax * Ddt[i]) * (Dlam[i] + amax * Ddlam[i]) : 0.0;
                    mu_aff = wave_reduce<OpSum>(s_aff) * inv_nineq;
                    double sigma = mu_aff * rcp_nr(mu); sigma = sigma * sigma * sigma;
                    if (alpha_prev < ADMPC_IPM_BLOCKED_STEP) sigma = 1.0;      // centring safeguard (admpc.h)
                    const double smu = fmax(sigma * mu, ADMPC_IPM_MU_FLOOR * tol_comp);      // admpc.h: centring target floor
                    if (!cons) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rc[i] = t[i] * lam[i] + dt[i] * dlam[i] - smu;
#pragma unroll
                        for (int i = 0; i < 2; ++i) Drc[i] = Dt[i] * Dlam[i] + Ddt[i] * Ddlam[i] - smu;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rc[i] = t[i] * lam[i] - smu;
#pragma unroll
                        for (int i = 0; i < 2; ++i) Drc[i] = Dt[i] * Dlam[i] - smu;
                    }
                } else {
                    double tau = 1.0 - mu_aff; tau = fmax(tau, 0.995); tau = fmin(tau, 0.999999);
                    const double alpha = fmin(tau * amax, 1.0);
                    if (it == 0 && warmed && alpha < wrest) {
                        // the first step from the warm start is blocked (cfg.ipm_warm_restart): start over from the cold start; the
                        // iteration counts.  Wave-uniform (alpha is).
                        warmed = false;
                        cold_start();
                    } else {
                    alpha_prev = alpha;
                    stp_local = uact ? fabs(alpha * ddu) : 0.0;
                    // idle lanes carry harmless finite values (their steps are computed from finite data)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { t[i] = fmax(t[i] + alpha * dt[i], IPM_FLOOR); lam[i] = fmax(lam[i] + alpha * dlam[i], IPM_FLOOR); }
                    du += alpha * ddu; sl += alpha * dsl; su += alpha * dsu;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        Dt[i] = dact ? fmax(Dt[i] + alpha * Ddt[i], IPM_FLOOR) : 1.0;
                        Dlam[i] = dact ? fmax(Dlam[i] + alpha * Ddlam[i], IPM_FLOOR) : 1.0;
                    }
                    dx6 += dact ? alpha * ddx6 : 0.0;
                    }
                }
                WSYNC();
            }
        }
        PHASE_STAMP(5);
        // ---------------- hand the step over to the expand kernel (H6): du per input, preliminary status, iteration count ----------------
        auxg_out[(size_t)inst * 128 + lane] = uact ? du : 0.0;          // the g0 slot of this instance is dead by now
        if (lane == 0) {
            statusg[inst] = failed ? ADMPC_STATUS_QP_FAILURE : ADMPC_STATUS_SUCCESS;
            if (failed && costg) costg[inst] = INFINITY;
            if (itersg) itersg[inst] = it;
#ifdef ADMPC_TRACE_SCHED
            { unsigned long long dbg_t1; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dbg_t1));
              unsigned long long dur = dbg_t1 - dbg_t0; if (dur > 32767) dur = 32767;          // 100 MHz ticks
              const unsigned long long pk = ((dbg_t0 & 0xfffffull) << 33) | (dur << 18) | ((unsigned long long)blockIdx.x << 6) | (unsigned long long)(it & 63);
              if (costg) costg[inst] = (double)pk; }
#endif
        }
        WSYNC();
        PHASE_STAMP(6);
    }
    PHASE_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// kernel E (N = 20 path): H6 -- expand the states through the linearised dynamics, full step, cost, status.
// One instance per wavefront.  Lane r < 7 owns state component r: dx_{k+1}[r] = b_k[r] + sum_c [A_k B_k][r][c] (dx_k, du_k)[c],
// the dx_k[c] picked up inside the FMAs by DPP row broadcasts (v_fmac_f64_dpp row_newbcast:c) -- no cross-lane exchange, no
// wait in the 20-stage chain; everything the chain needs from LDS is independent of it and prefetched by the compiler.
// A kernel of its own: inside the interior-point kernel (256 registers, 2 waves/SIMD) the same work ran serialised on
// single loads and took as long as six factorisations.
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(WAVE) void admpc_expand_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                            const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                            const double* __restrict__ yrefeg,
                                                            const double* __restrict__ GTg, const double* __restrict__ blg,
                                                            double* __restrict__ xbarg, double* __restrict__ ubarg,
                                                            double* __restrict__ costg, int32_t* __restrict__ statusg,
                                                            const double* __restrict__ dug)
{
    constexpr int N = NT, n = 2 * NT;
    extern __shared__ double lds_raw[];
    double* const GT = lds_raw;                             // packed linearisation
    double* const bl = GT + N * GTS;                        // defects b_k
    double* const dq = bl + DenseLds<NT>::BLS;              // xbar_k - xref_k, k = 0..N (overwritten by dx_k)
    double* const dus = dq + DenseLds<NT>::DQS;             // [64] du per input
    const int lane = threadIdx.x;
    const int ji = lane & 1;
    const bool uact = lane < n;
    const int r6 = lane < 6 ? lane : 0;                     // row of the packed linearisation this lane reads
    const int r7 = lane < NX ? lane : 0;
    const double Ts = cfg->Ts, h = cfg->Ts;
    const double wq = lane < NX ? Ts * cfg->W[r7] : 0.0, wqe = lane < NX ? cfg->We[r7] : 0.0;
    const double Rj = Ts * cfg->W[NX + ji];
    const double rho_l = Ts * cfg->zl, rho_u = Ts * cfg->zu;
    for (int inst = blockIdx.x; inst < B; inst += gridDim.x) {
        if (statusg[inst] != 0) continue;                   // failed in the interior-point kernel (or failed / converged in an earlier SQP iteration)
        const double* xbg = xbarg + (size_t)inst * (N + 1) * NX;
        const double* ubg = ubarg + (size_t)inst * N * NU;
        const double* yrg = yrefg + (size_t)inst * N * NY;
        stage_in<N * GTS>(GT, GTg + (size_t)inst * N * GTS, lane);
        stage_in<N * NX>(bl, blg + (size_t)inst * N * NX, lane);
        stage_dq<N>(dq, xbg, yrg, yrefeg + (size_t)inst * NX, lane);
        const double du = dug[(size_t)inst * 128 + lane];  // 0 on idle lanes
        dus[lane] = du;
        const int sc = uact ? lane : 0;
        const double ubar_i = ubg[sc];
        const double uref_i = yrg[(sc >> 1) * 9 + 7 + (sc & 1)];
        double dx = lane < NX ? x0g[(size_t)inst * NX + r7] - xbg[r7] : 0.0;      // dx_0 (lanes 0..6)
        WSYNC();
        bool bad = false;
        double J = 0.0;
        static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double e = dx + dq[k * 7 + r7];
            J += 0.5 * (k < N ? wq : wqe) * e * e;
            if (!(fabs(dx) <= 1e300)) bad = true;
            if (lane < NX) dq[k * 7 + lane] = dx;            // slot k now holds dx_k
            if constexpr (k < N) {
                const double* Gk = GT + k * GTS;
                const double u0 = dus[2 * k], u1 = dus[2 * k + 1];
                // rows 0..5: b + [e0 e1 A(:,2..6)] dx + B du ; row 6: delta' = delta + h u1
                double acc = bl[k * 7 + r7] + (lane < 2 || lane == 6 ? dx : 0.0);
                double g[5];
#pragma unroll
                for (int c = 0; c < 5; ++c) g[c] = lane < 6 ? Gk[c * 6 + r6] : 0.0;
                const double b0 = lane < 6 ? Gk[5 * 6 + r6] : 0.0, b1 = lane < 6 ? Gk[6 * 6 + r6] : (lane == 6 ? h : 0.0);
                acc += b0 * u0 + b1 * u1;
                fmac_rowbc<2>(acc, dx, g[0]); fmac_rowbc<3>(acc, dx, g[1]); fmac_rowbc<4>(acc, dx, g[2]);
                fmac_rowbc<5>(acc, dx, g[3]); fmac_rowbc<6>(acc, dx, g[4]);
                dx = lane < NX ? acc : 0.0;
            }
        });
        const double unew = ubar_i + du;
        if (uact && !(fabs(unew) <= 1e300)) bad = true;
        const int status = __any(bad) ? ADMPC_STATUS_QP_FAILURE : ADMPC_STATUS_SUCCESS;
        double Ju = 0.0;
        WSYNC();
        if (status == 0) {
            double* xo = xbarg + (size_t)inst * (N + 1) * NX;
            double* uo = ubarg + (size_t)inst * N * NU;
#pragma unroll
            for (int i0 = 0; i0 < (N + 1) * NX; i0 += WAVE) { const int i = i0 + lane; if (i < (N + 1) * NX) xo[i] = xbg[i] + dq[i]; }
            if (uact) {
                const double e = unew - uref_i;
                Ju = 0.5 * Rj * e * e;
                if (unew < cfg->lbu[ji]) Ju += rho_l * (cfg->lbu[ji] - unew);
                if (unew > cfg->ubu[ji]) Ju += rho_u * (unew - cfg->ubu[ji]);
                uo[lane] = unew;
            }
        }
        const double Jt = wave_reduce<OpSum>(J + Ju);
        if (lane == 0) {
#ifndef ADMPC_TRACE_SCHED
            if (costg) costg[inst] = status == 0 ? Jt : INFINITY;
#endif
            statusg[inst] = status;
        }
        WSYNC();
    }
}
#endif   // ADMPC_LEGACY_N20

// ---------------------------------------------------------------------------------------------
// local reference generator: batched RefTrajectory.get_waypoints (src/ad_mpc/ref_traj.py:89-171).
// One wavefront per vehicle pose; the global trajectory (M waypoints: vel, x, y, psi, unwrapped psi, cdist, curv)
// is shared.  Lane h < H owns horizon slot h.  Reproduces the reference's arithmetic, including that the
// interpolation abscissae start at the beginning of the path (start_dist is computed but unused, :125-134).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double bound_pi(double a) {            // (a + pi) % (2 pi) - pi with Python's modulo (ref_traj.py:29-30)
    const double twopi = 2.0 * M_PI;
    double r = fmod(a + M_PI, twopi);
    if (r != 0.0 && r < 0.0) r += twopi;
    return r - M_PI;
}
__device__ __forceinline__ double interp_np(const double* __restrict__ xp, const double* __restrict__ fp, int M, double x) {   // numpy.interp
    if (x <= xp[0]) return fp[0];
    if (x >= xp[M - 1]) return fp[M - 1];
    int lo = 0, hi = M - 1;                                       // xp[lo] <= x < xp[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
    const double slope = __ddiv_rn(__dsub_rn(fp[lo + 1], fp[lo]), __dsub_rn(xp[lo + 1], xp[lo]));
    return __dadd_rn(__dmul_rn(slope, __dsub_rn(x, xp[lo])), fp[lo]);
}

__global__ __launch_bounds__(WAVE) void admpc_waypoints_kernel(int M, int H, double dt, int B,
        const double* __restrict__ vel, const double* __restrict__ tx, const double* __restrict__ ty,
        const double* __restrict__ tpsi, const double* __restrict__ tpsi_unw, const double* __restrict__ cdist, const double* __restrict__ curv,
        const double* __restrict__ Xi, const double* __restrict__ Yi, const double* __restrict__ Pi,
        double* __restrict__ out_ref /*[B][6][H]: x,y,psi,v,cdist,curv*/, double* __restrict__ out_err /*[B][3]: s0,e_y0,e_psi0*/,
        int32_t* __restrict__ out_stop)
{
    __shared__ double sh[WAVE];
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const double X0 = Xi[b], Y0 = Yi[b];
        const double psi0 = bound_pi(Pi[b]);
        // (1) closest waypoint: first index of the minimum of sqrt(dx^2 + dy^2)
        double best = INFINITY; int bi = 0x7fffffff;
        for (int m = lane; m < M; m += WAVE) {
            const double dx = __dsub_rn(tx[m], X0), dy = __dsub_rn(ty[m], Y0);
            const double d = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
            if (d < best) { best = d; bi = m; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, WAVE); const int oi = __shfl_xor(bi, o, WAVE);
            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const int ci = bi;
        // (2) Frenet errors at the closest waypoint
        if (lane == 0) {
            const double pw = tpsi[ci];
            const double ex = __dsub_rn(X0, tx[ci]), ey = __dsub_rn(Y0, ty[ci]);
            out_err[b * 3 + 0] = cdist[ci];
            out_err[b * 3 + 1] = __dadd_rn(__dmul_rn(-sin(pw), ex), __dmul_rn(cos(pw), ey));
            out_err[b * 3 + 2] = bound_pi(psi0 - pw);
        }
        // (3) abscissae: cumulative dt * vel over the horizon (velocities padded with 0.01), lane h -> s_h
        double s_h = 0.0;
        {
            double acc = 0.0;
            for (int h = 0; h < H; ++h) {                 // serial, identical rounding to the reference's running sum
                const double v = h < M ? vel[h] : 0.01;
                acc = h == 0 ? __dmul_rn(dt, v) : __dadd_rn(acc, __dmul_rn(dt, v));
                if (h == lane) s_h = acc;
            }
        }
        const bool on = lane < H;
        const double xr = on ? interp_np(cdist, tx, M, s_h) : 0.0;
        const double yr = on ? interp_np(cdist, ty, M, s_h) : 0.0;
        const double cr = on ? interp_np(cdist, cdist, M, s_h) : 0.0;
        const double kr = on ? interp_np(cdist, curv, M, s_h) : 0.0;
        const double pr = on ? interp_np(cdist, tpsi_unw, M, s_h) : 0.0;
        // psi: fix_angle_reference (bound, unwrap, add back) then bound (ref_traj.py:32-37,146-148)
        const double d0 = bound_pi(pr - psi0);
        sh[lane] = d0;
        __syncthreads();
        double corr = 0.0;
        if (on && lane >= 1) {
            const double dd = __dsub_rn(d0, sh[lane - 1]);
            double ddmod = fmod(dd + M_PI, 2.0 * M_PI);
            if (ddmod < 0.0) ddmod += 2.0 * M_PI;
            ddmod -= M_PI;
            if (ddmod == -M_PI && dd > 0.0) ddmod = M_PI;
            corr = fabs(dd) < M_PI ? 0.0 : __dsub_rn(ddmod, dd);
        }
        __syncthreads();
        // cumulative sum of the corrections (serial order as numpy.cumsum)
        sh[lane] = corr;
        __syncthreads();
        double cum = 0.0;
        for (int h = 1; h <= lane && h < H; ++h) cum = __dadd_rn(cum, sh[h]);
        const double psi_fixed = bound_pi(__dadd_rn(psi0, lane >= 1 ? __dadd_rn(d0, cum) : d0));
        __syncthreads();
        // v_ref = diff(cdist_ref) / dt, last value repeated
        sh[lane] = cr;
        __syncthreads();
        double vr = 0.0;
        if (on) {
            const int h1 = lane < H - 1 ? lane : H - 2;
            vr = __ddiv_rn(__dsub_rn(sh[h1 + 1], sh[h1]), dt);
        }
        if (lane == 0) out_stop[b] = (sh[H - 1] == cdist[M - 1]) ? 1 : 0;
        __syncthreads();
        // splice: three points from the current pose to the second waypoint, then waypoints 2..H-2 (ref_traj.py:158-170)
        double* o = out_ref + (size_t)b * 6 * H;
        sh[lane] = xr; __syncthreads();
        const double x1 = sh[1];
        double xo = 0.0;
        if (on) { if (lane < 3) { const double st = __ddiv_rn(__dsub_rn(x1, X0), 2.0); xo = lane == 2 ? x1 : __dadd_rn(X0, __dmul_rn((double)lane, st)); } else xo = sh[lane - 1]; }
        __syncthreads();
        sh[lane] = yr; __syncthreads();
        const double y1 = sh[1];
        double yo = 0.0;
        if (on) { if (lane < 3) { const double st = __ddiv_rn(__dsub_rn(y1, Y0), 2.0); yo = lane == 2 ? y1 : __dadd_rn(Y0, __dmul_rn((double)lane, st)); } else yo = sh[lane - 1]; }
        __syncthreads();
        sh[lane] = psi_fixed; __syncthreads();
        const double po = on ? (lane < 3 ? sh[0] : sh[lane - 1]) : 0.0;
        __syncthreads();
        sh[lane] = vr; __syncthreads();
        const double vo = on ? (lane < 3 ? sh[2] : sh[lane - 1]) : 0.0;
        __syncthreads();
        if (on) { o[0 * H + lane] = xo; o[1 * H + lane] = yo; o[2 * H + lane] = po; o[3 * H + lane] = vo; o[4 * H + lane] = cr; o[5 * H + lane] = kr; }
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
    if (threadIdx.x >= LIN_TASKS) return;                // same task -> lane map as the linearisation kernel (gp_eval relies on it)
    for (long tsk = (long)blockIdx.x * LIN_TASKS + threadIdx.x; tsk < total; tsk += (long)gridDim.x * LIN_TASKS) {
        const long sk = tsk / 3; const int g = (int)(tsk % 3);
        const long inst = sk / N; const int k = (int)(sk % N);
        double x[NX], u[NU], phi[NX], col[3][NX];
        for (int i = 0; i < NX; ++i) x[i] = xbarg[(inst * (N + 1) + k) * NX + i];
        u[0] = ubarg[(inst * N + k) * NU]; u[1] = ubarg[(inst * N + k) * NU + 1];
        rk4_group<double>(cfg, x, u, pg[inst], cfg->Ts, g, phi, col);
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

// receding-horizon shift of the iterate: stage k takes the values of stage k+1, the last input is kept and the new
// terminal state is either a copy (rollout = 0) or one model step from the old terminal state under the last input.
// Three lanes per instance as in the linearisation kernel (gp_eval shares its sums among them); lane 0 moves x, lane 1 moves u.
__global__ __launch_bounds__(WAVE) void admpc_shift_kernel(const AdmpcConfig* __restrict__ cfg, int B, double* __restrict__ xbarg,
                                                           double* __restrict__ ubarg, const double* __restrict__ pg, int rollout)
{
    const int N = cfg->N;
    const long total = (long)B * 3;
    if (threadIdx.x >= LIN_TASKS) return;
    for (long tsk = (long)blockIdx.x * LIN_TASKS + threadIdx.x; tsk < total; tsk += (long)gridDim.x * LIN_TASKS) {
        const long inst = tsk / 3; const int g = (int)(tsk % 3);
        double* xb = xbarg + inst * (N + 1) * NX;
        double* ub = ubarg + inst * N * NU;
        double x[NX], u[NU], phi[NX], col[3][NX];
        for (int i = 0; i < NX; ++i) { x[i] = xb[N * NX + i]; phi[i] = x[i]; }
        u[0] = ub[(N - 1) * NU]; u[1] = ub[(N - 1) * NU + 1];
        if (rollout) rk4_group<double>(cfg, x, u, pg[inst], cfg->Ts, g, phi, col);
        if (g == 0) {
            for (int i = 0; i < N * NX; ++i) xb[i] = xb[i + NX];
            for (int i = 0; i < NX; ++i) xb[N * NX + i] = phi[i];
        } else if (g == 1) {
            for (int i = 0; i < (N - 1) * NU; ++i) ub[i] = ub[i + NU];
        }
    }
}

// arg-min over cost[0..B): one block; the ordering rules (ties -> lowest index; NaN read as +inf) live in argmin_rule.h
__global__ __launch_bounds__(256) void admpc_argmin_kernel(const double* __restrict__ cost, int B, int64_t offset,
                                                           double* __restrict__ val, int64_t* __restrict__ idx)
{
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    ArgminBest b = argmin_identity();
    for (int i = threadIdx.x; i < B; i += blockDim.x) argmin_fold(b, argmin_cost(cost[i]), (int64_t)i);
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(b.v, o, WAVE);
        const int64_t oi = __shfl_xor((long long)b.i, o, WAVE);
        argmin_fold(b, ov, oi);
    }
    const int w = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0) { sv[w] = b.v; si[w] = b.i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) argmin_fold(b, sv[i], si[i]);
        *val = b.v;
        *idx = argmin_final_index(b) + offset;
    }
}

// second level of the arg-min: W gathered (cost, global index) pairs, 16 bytes each, as the per-GPU admpc_argmin wrote them
// and an all-gather laid them out; one wave; same rules (argmin_rule.h)
__global__ __launch_bounds__(WAVE) void admpc_argmin_pairs_kernel(const double* __restrict__ pairs, int W,
                                                                  double* __restrict__ val, int64_t* __restrict__ idx)
{
    ArgminBest b = argmin_identity();
    for (int i = threadIdx.x; i < W; i += WAVE)
        argmin_fold(b, argmin_cost(pairs[2 * i]), (int64_t)__double_as_longlong(pairs[2 * i + 1]));
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(b.v, o, WAVE);
        const int64_t oi = __shfl_xor((long long)b.i, o, WAVE);
        argmin_fold(b, ov, oi);
    }
    if (threadIdx.x == 0) { *val = b.v; *idx = argmin_final_index(b); }
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

// SQP solve with a tolerance (cfg.sqp_iters > 1, cfg.sqp_tol > 0: reference solver_type "SQP", create_ros_ad_mpc.py:47-51; the tolerances
// are acados' defaults nlp_solver_tol_{stat,eq,ineq,comp} = 1e-6, acados_models/sim_car_acados_ocp.json:870-873): acados' stopping test
// (ocp_nlp_sqp.c: linearise -> residuals of the NLP's KKT system with the iterate's multipliers -> all four inf-norms within tolerance:
// ACADOS_SUCCESS, else solve the QP and step; ACADOS_MAXITER after nlp_solver_max_iter QPs).  Runs between the linearisation and kernel R
// in every pass of a solve but the first (the C ABI takes no multipliers in: a cold solver), on the NEW linearisation GT / bl with the
// multipliers kernel R left for the previous QP (pi: adjoint recursion, as HPIPM's expansion of the condensed solution; ineq: slacks t and
// multipliers lam in the record order of admpc.h):
//   res_stat  rows of grad L:  u: R (u - uref) + B' pi_k - lam_lo + lam_up;  slacks: rho - lam - lam_s;
//                              x_k: Q (x_k - xref_k) + A_k' pi_k - pi_{k-1} (- lam_d,lo + lam_d,up on delta);  x_N: Q_e (x_N - xref_e) - pi_{N-1};
//                              x_0: against pi_N, the multiplier of the initial-state equality
//   res_eq    shooting defects b_k, x0 - x_0
//   res_ineq  constraint value minus its slack t (slack variables read from the slacks of their own bounds)
//   res_comp  lam .* t
// One wave per instance, lane <-> stage.  A converged instance gets status -1 and is skipped by the rest of the solve.
// fp32 (kernel R's float instantiation stops its QPs at residual 1e-2 / complementarity 1e-3, rowqp_core.h): the tolerances are floored
// at those levels (and the defects at 1e-4: the float shooting), since no NLP residual can be driven below the QP's own.
template <class T>
__global__ __launch_bounds__(WAVE) void admpc_nlp_res_kernel(const AdmpcConfig* __restrict__ cfg, int B, const T* __restrict__ x0g,
                                                             const T* __restrict__ yrefg, const T* __restrict__ yrefeg,
                                                             const T* __restrict__ xbarg, const T* __restrict__ ubarg,
                                                             const T* __restrict__ GTg, const T* __restrict__ blg,
                                                             const T* __restrict__ pig, const T* __restrict__ ineqg,
                                                             int32_t* __restrict__ statusg, T* __restrict__ resg)
{
    constexpr bool f32 = sizeof(T) == 4;
    const int N = cfg->N, lane = threadIdx.x;
    const T h = (T)cfg->Ts;
    const T rho_l = (T)(cfg->Ts * cfg->zl), rho_u = (T)(cfg->Ts * cfg->zu);
    const double tol = cfg->sqp_tol;
    const double tol_stat = f32 && tol < 1e-2 ? 1e-2 : tol, tol_eq = f32 && tol < 1e-4 ? 1e-4 : tol;
    const double tol_ineq = tol_stat, tol_comp = f32 && tol < 1e-3 ? 1e-3 : tol;
    for (int inst = blockIdx.x; inst < B; inst += gridDim.x) {
        if (statusg[inst] != 0) continue;                                   // failed or converged in an earlier pass
        const T* xb = xbarg + (size_t)inst * (N + 1) * NX;
        const T* ub = ubarg + (size_t)inst * N * NU;
        const T* yr = yrefg + (size_t)inst * N * NY;
        const T* pi = pig + (size_t)inst * (N + 1) * NX;
        T rs = 0, re = 0, ri = 0, rc = 0;
        auto upd = [](T& acc, T v) __attribute__((always_inline)) { const T a = v < 0 ? -v : v; if (a > acc || a != a) acc = a; };
        for (int k = lane; k <= N; k += WAVE) {
            T x[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xb[k * NX + i];
            if (k == N) {
#pragma unroll
                for (int i = 0; i < NX; ++i) upd(rs, (T)cfg->We[i] * (x[i] - yrefeg[(size_t)inst * NX + i]) - pi[(N - 1) * NX + i]);
                continue;
            }
            const T* G = GTg + ((size_t)inst * N + k) * GTS;
            const T* pk = pi + k * NX;
            const T* pp = pi + (k >= 1 ? k - 1 : N) * NX;
            const T* iq = ineqg + ((size_t)inst * N + k) * 20;
            T pv[NX], t[10], lm[10];
#pragma unroll
            for (int i = 0; i < NX; ++i) pv[i] = pk[i];
#pragma unroll
            for (int i = 0; i < 10; ++i) { t[i] = iq[i]; lm[i] = iq[10 + i]; }
            // stationarity in x_k: columns 0, 1 of A_k are unit vectors, row 6 of [A B] is [e6, 0, h] (not stored)
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T a = (T)(cfg->Ts * cfg->W[i]) * (x[i] - yr[k * NY + i]) - pp[i];
                if (i < 2) a += pv[i];
                else {
#pragma unroll
                    for (int r = 0; r < 6; ++r) a += G[(i - 2) * 6 + r] * pv[r];
                    if (i == 6) { a += pv[6]; if (k >= 1) a += lm[5] - lm[4]; }
                }
                upd(rs, a);
            }
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const T u = ub[k * NU + j];
                T a = (T)(cfg->Ts * cfg->W[NX + j]) * (u - yr[k * NY + NX + j]) - lm[2 * j] + lm[2 * j + 1];
#pragma unroll
                for (int r = 0; r < 6; ++r) a += G[(5 + j) * 6 + r] * pv[r];
                if (j == 1) a += h * pv[6];
                upd(rs, a);
                upd(rs, rho_l - lm[2 * j] - lm[6 + 2 * j]); upd(rs, rho_u - lm[2 * j + 1] - lm[7 + 2 * j]);
                upd(ri, u + t[6 + 2 * j] - (T)cfg->lbu[j] - t[2 * j]); upd(ri, (T)cfg->ubu[j] - u + t[7 + 2 * j] - t[2 * j + 1]);
                upd(rc, lm[2 * j] * t[2 * j]); upd(rc, lm[2 * j + 1] * t[2 * j + 1]);
                upd(rc, lm[6 + 2 * j] * t[6 + 2 * j]); upd(rc, lm[7 + 2 * j] * t[7 + 2 * j]);
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) upd(re, blg[((size_t)inst * N + k) * NX + i]);
            if (k >= 1) {
                upd(ri, x[6] - (T)cfg->lbx_delta - t[4]); upd(ri, (T)cfg->ubx_delta - x[6] - t[5]);
                upd(rc, lm[4] * t[4]); upd(rc, lm[5] * t[5]);
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i) upd(re, x0g[(size_t)inst * NX + i] - x[i]);
            }
        }
        const double ws = wave_reduce<OpMaxNan>((double)rs), we = wave_reduce<OpMaxNan>((double)re);
        const double wi = wave_reduce<OpMaxNan>((double)ri), wc = wave_reduce<OpMaxNan>((double)rc);
        if (lane == 0) {
            if (ws <= tol_stat && we <= tol_eq && wi <= tol_ineq && wc <= tol_comp) statusg[inst] = -1;
            if (resg) { resg[(size_t)inst * 4 + 0] = (T)ws; resg[(size_t)inst * 4 + 1] = (T)we; resg[(size_t)inst * 4 + 2] = (T)wi; resg[(size_t)inst * 4 + 3] = (T)wc; }
        }
    }
}

// end of an SQP solve with a tolerance: -1 (converged in some step) -> 0, still 0 after the last step -> ADMPC_STATUS_MAXITER
__global__ void admpc_sqp_finalize_kernel(int B, int32_t* __restrict__ status)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int st = status[b];
    status[b] = st == -1 ? ADMPC_STATUS_SUCCESS : (st == 0 ? ADMPC_STATUS_MAXITER : st);
}

// Speed-reference clamp in front of the solve: gp_ad_mpc_node.py:344-349 resample_vel -- the reference speed of
// slot i may not exceed |v| + i * (acc_max * dt * 0.8), the bound growing by repeated addition as in the reference (same rounding).
// One thread per vehicle; vel_ref [B] rows of H values, `ld` values apart (e.g. row 3 of admpc_waypoints_batch's out_ref: ld = 6 H).
__global__ void admpc_resample_vel_kernel(int B, int H, int ld, const double* __restrict__ vx, const double* __restrict__ vy,
                                          double acc_max, double dt, double* __restrict__ vel_ref)
{
#pragma clang fp contract(off)      // every product and sum rounded on its own, as the host's Python arithmetic does (hipcc contracts by default)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // plain operators under the pragma above (the __dmul_rn / __dadd_rn wrappers of the HIP headers are inlined WITH the translation
    // unit's contraction flag and would be fused into an FMA)
    const double sx = vx[b] * vx[b], sy = vy[b] * vy[b];
    double bound = __dsqrt_rn(sx + sy);
    const double inc = (acc_max * dt) * 0.8;
    double* v = vel_ref + (size_t)b * ld;
    for (int i = 0; i < H; ++i) {
        if (v[i] > bound) v[i] = bound;
        bound = bound + inc;
    }
}

// Post-solve safety and actuation, one thread per vehicle slot / candidate:
//   check_pred_trj                          gp_ad_mpc_node.py:248-257 (same formula as is_valid_command, ad_3d_optimizer.py:385-394)
//   consecutive-success gate                :206-213  (status > 0 resets the counter; fewer than `threshold` successes: no MPC command)
//   steering command                        :222-223  clip(clip(rate) * 0.1 + measured steering)
//   fallback (auxiliary controller)         :455-476  steering held at the measured value, acceleration -1e5 (hard brake)
//   Ackermann record                        create_ros_ad_mpc.py:95-98 (float32 message fields)
// cost_io (may be null): +inf for every candidate that does not produce an MPC command, so that an arg-min over it picks
// a valid candidate only.
__global__ void admpc_actuation_kernel(int N, int B, const double* __restrict__ xopt, const double* __restrict__ uopt,
                                       const double* __restrict__ xref_xy, const int32_t* __restrict__ status,
                                       const double* __restrict__ steer_meas, int32_t* __restrict__ safe_count, int threshold,
                                       double rate_min, double rate_max, double steer_min, double steer_max,
                                       double* __restrict__ cost_io, float* __restrict__ ack, int32_t* __restrict__ mode, int32_t* __restrict__ valid)
{
#pragma clang fp contract(off)      // the steering command is two separately rounded operations in the reference (:223)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* x = xopt + (size_t)b * (N + 1) * NX;
    const double* r = xref_xy + (size_t)b * (N + 1) * 2;
    const int n = N + 1;
    double sm = 0.0, mx = 0.0;
    for (int i = 0; i < n - 1; ++i) {
        const double dxv = r[i * 2] - x[i * 7], dyv = r[i * 2 + 1] - x[i * 7 + 1];
        const double d = sqrt(dxv * dxv + dyv * dyv);
        sm += d; mx = fmax(mx, d);
    }
    const double mean = sm / n;
    double var = 0.0;
    for (int i = 0; i < n; ++i) {
        double d = 0.0;
        if (i < n - 1) { const double dxv = r[i * 2] - x[i * 7], dyv = r[i * 2 + 1] - x[i * 7 + 1]; d = sqrt(dxv * dxv + dyv * dyv); }
        var += (d - mean) * (d - mean);
    }
    var /= (n - 1);
    const int healthy = (mean < 3.0 && var < 2.0 && mx < 4.0) ? 1 : 0;
    const int cnt = status[b] > 0 ? 0 : safe_count[b] + 1;
    safe_count[b] = cnt;
    const int ok = (cnt >= threshold && healthy) ? 1 : 0;
    const double* u = uopt + (size_t)b * N * NU;
    const double sth = steer_meas[b];
    if (ok) {
        const double rate_msg = (double)(float)u[1];                 // the value travels through a float32 message field
        const double sv = fmax(fmin(rate_max, rate_msg), rate_min);
        const double scaled = sv * 0.1;
        const double ang = fmax(fmin(steer_max, scaled + sth), steer_min);
        ack[b * 4 + 0] = (float)ang; ack[b * 4 + 1] = (float)u[1]; ack[b * 4 + 2] = (float)x[3]; ack[b * 4 + 3] = (float)u[0];
    } else {
        ack[b * 4 + 0] = (float)sth; ack[b * 4 + 1] = 0.0f; ack[b * 4 + 2] = 0.0f; ack[b * 4 + 3] = (float)(-1e5);
    }
    mode[b] = ok;
    valid[b] = healthy;
    if (cost_io && !ok) cost_io[b] = INFINITY;
}

}  // namespace

// =============================================================================================
// C ABI (include/admpc.h)
// =============================================================================================
#include <string>
#include <dlfcn.h>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <cstdlib>

struct AdmpcSolver {
    AdmpcConfig cfg;
    AdmpcConfig* d_cfg;
    int device;
    int num_cu;
    int use_dense;           // condensed dense-Cholesky QP kernel available for this horizon (N == 20) and not disabled
    int dense_lds_bytes;
    int n20_fused;           // N = 20 fp64 steps run the fused persistent kernel (admpc_fused20.hip); 0: the four-kernel pipeline (ADMPC_N20=split)
    int use_seg;             // N = 40 / 60 / 80 fp64 steps without GP models run the segmented condensed kernel (admpc_seg.hip; with GP models on ADMPC_QP=seg); ADMPC_QP=riccati: kernel R
    int* d_tick;             // 2 x [128 + 64 cap_fused] tickets, exit counter and work-order bins of the persistent kernels: two states used alternately (work_order.h)
    int tick_flip;           // which of the two the last launch used
    int cap_fused;
    double* d_slot;          // per-wave slot buffers of the fused kernel (the linearisation across the interior point), allocated at its first launch
    // Workspaces, each grown on demand by the path that needs it (admpc_reserve sizes the handle's default path up front):
    int cap;                 // instances: d_status
    int32_t* d_status;       // [cap] used when the caller passes status == NULL
    int cap_lin;             // kernel A's output: every path but the fused one
    double* d_GT;            // [cap_lin][N][42]
    double* d_bl;            // [cap_lin][N][7]
    int cap_dense;           // four-kernel N = 20 pipeline
    double* d_H;             // [cap_dense][NTRI] condensed Hessians
    double* d_aux;           // [cap_dense][128]
    int* d_sched;            // [SCHED_HDR (+ SCHED_NB * cap_dense)] ticket counter of kernel R / work scheduler of the four-kernel pipeline
    int sched_cap;           // instances the scheduler lists of d_sched were sized for (0: header only)
    int qmask;               // 7 when only x, y, psi carry tracking weights (specialised condensing kernel), else 127
    int cap_row;             // kernel R: every fp32 solve, fp64 for N != 20, every solve that asks for multipliers
    int row_elem;            // element width (8 / 4) the row workspace was sized for
    double* d_ws;            // [cap_row][N+1][38] workspace of the row kernel (sweep-private state, L2-resident)
    int32_t* d_split;        // [2 cap_row + 1] keys, order and count of the row kernel's second phase (split batches)
    double* d_dump;          // [cap_row][16 + 31 N] LDS regions of the deferred instances between the two phases
    double* d_mult;          // [cap_mult][(N+1) 7 + 20 N] multipliers between the passes of an SQP solve with a tolerance, when the caller keeps none
    int cap_mult, mult_elem;
    int row_chunk;           // > 0: cap of the chunk size of kernel-R solves (ADMPC_ROWQP_CHUNK; tests)
    int split_mode;          // -1: split batches of more than one round of waves (default), 0: never, 1: always (ADMPC_ROWQP_SPLIT)
    double* d_pairs;         // [1 + 256] 16-byte (cost, index) records: this rank's, then the all-gathered ones (admpc_argmin_global)
};

// Every entry point runs on the solver's device and restores the caller's current device on return (a host with several GPUs
// keeps its own current device: torch allocations and default-stream work of the caller are not redirected).
struct DeviceGuard {
    int prev; bool switched; bool good;
    explicit DeviceGuard(int dev) : prev(-1), switched(false), good(true) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { good = hipSetDevice(dev) == hipSuccess; switched = good; }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    bool ok() const { return good; }
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(ADMPC_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// kernel R (admpc_rowqp.hip)
extern "C" int admpc_rowqp_plan(int N, int elem, int B, int num_cu, int* rows, int* inst_stride, int* lds_bytes, int* grid);
extern "C" void admpc_rowqp_prepare(void);
extern "C" void admpc_rowqp_launch_f64(int grid, int lds_bytes, hipStream_t st, const AdmpcConfig* d_cfg, int B, int rows, int inst_stride,
        const double* x0, const double* yref, const double* yref_e, const double* GT, const double* bl,
        double* xbar, double* ubar, double* cost, int32_t* stat, int32_t* iters, double* pi, double* ineq, double* ws, int first, int* ticket, int32_t* split, double* dump);
extern "C" void admpc_rowqp_launch_f32(int grid, int lds_bytes, hipStream_t st, const AdmpcConfig* d_cfg, int B, int rows, int inst_stride,
        const float* x0, const float* yref, const float* yref_e, const float* GT, const float* bl,
        float* xbar, float* ubar, float* cost, int32_t* stat, int32_t* iters, float* pi, float* ineq, float* ws, int first, int* ticket, int32_t* split, float* dump);

// fused N = 20 step (admpc_fused20.hip)
extern "C" void admpc_fused20_launch(int num_cu, hipStream_t st, const AdmpcConfig* d_cfg, int B, int qmask,
        const double* x0, const double* yref, const double* yref_e, const double* p, double* xbar, double* ubar,
        double* cost, int32_t* stat, int32_t* iters, int first, int* sched2, int cap, int flip, double* slotbuf);
extern "C" size_t admpc_fused20_slot_doubles(int num_cu);
extern "C" size_t admpc_fused20_sched_ints(int cap);
// segmented condensed step, N = 40 / 60 / 80 fp64 (admpc_seg.hip)
extern "C" int admpc_seg_supports(int N);
extern "C" void admpc_seg_launch(int N, int num_cu, hipStream_t st, const AdmpcConfig* d_cfg, int B, int qmask,
        const double* x0, const double* yref, const double* yref_e, const double* p, double* xbar, double* ubar,
        double* cost, int32_t* stat, int32_t* iters, int first, int* sched2, int cap, int flip, double* hslot);
extern "C" size_t admpc_seg_slot_doubles(int num_cu);

extern "C" {

const char* admpc_last_error(void) { return g_err.c_str(); }
// for the other translation units of the library (admpc_quad.hip)
__attribute__((visibility("hidden"))) int admpc_set_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }
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
    c->ipm_mu0 = 1.0; c->ipm_thr0 = 0.1; c->ipm_tol_comp = 1e-8; c->ipm_tol_res = 1e-8; c->ipm_tol_step = 1e30;      // HPIPM BALANCE's levels (the reference's setting), no step test; admpc.h
    c->ipm_try_unconstrained = 1.0; c->ipm_warm_thr = 0.01; c->ipm_warm_restart = 0.1; c->ipm_fallback_iter = 30.0;
    return ADMPC_OK;
}

static int validate(const AdmpcConfig* c)
{
    if (c->N < 2 || c->N > ADMPC_MAX_N) return fail(ADMPC_EINVAL, "N out of range [2,128]");
    if (!(c->Ts > 0)) return fail(ADMPC_EINVAL, "Ts must be positive");
    if (c->n_gp < 0 || c->n_gp > ADMPC_GP_MAX) return fail(ADMPC_EINVAL, "n_gp out of range");
    for (int g = 0; g < c->n_gp; ++g) {
        const AdmpcGp& gp = c->gp[g];
        bool okf = gp.n_feat >= 1 && gp.n_feat <= ADMPC_GP_MAX_FEAT;
        for (int d = 0; okf && d < gp.n_feat; ++d) okf = gp.feat[d] >= 3 && gp.feat[d] <= 8;
        if (gp.out < 3 || gp.out > 5 || !okf || gp.n_points < 0 || gp.n_points > ADMPC_GP_MAX_POINTS)
            return fail(ADMPC_EINVAL, "GP: out must be in {3,4,5}, 1..3 features in {3..8}, n_points <= 32");
    }
    if (!(c->W[NX] > 0 && c->W[NX + 1] > 0)) return fail(ADMPC_EINVAL, "input weights must be positive (strict convexity)");
    if (c->ipm_iter_max < 1) return fail(ADMPC_EINVAL, "ipm_iter_max < 1");
    if (!(c->sqp_tol >= 0)) return fail(ADMPC_EINVAL, "sqp_tol must be >= 0");
    if (!(c->ipm_mu0 > 0) || !(c->ipm_thr0 > 0) || !(c->ipm_warm_thr >= 0) || !(c->ipm_warm_restart >= 0 && c->ipm_warm_restart < 1) || !(c->ipm_fallback_iter >= 0 && c->ipm_fallback_iter <= 1e6)) return fail(ADMPC_EINVAL, "ipm_mu0, ipm_thr0 must be > 0, ipm_warm_thr >= 0, ipm_warm_restart in [0, 1), ipm_fallback_iter in [0, 1e6]");
    return ADMPC_OK;
}

int admpc_create(const AdmpcConfig* cfg, int device, AdmpcSolver** out)
{
    if (!cfg || !out) return fail(ADMPC_EINVAL, "admpc_create: null argument");
    int rc = validate(cfg); if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ADMPC_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return fail(ADMPC_ENODEV, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    AdmpcSolver* s = new (std::nothrow) AdmpcSolver();
    if (!s) return fail(ADMPC_ENOMEM, "out of host memory");
    s->cfg = *cfg; s->device = device; s->num_cu = prop.multiProcessorCount;
    {   // the row kernel must fit at least one instance per wave into the 160 KB of LDS
        int r_, st_, lb_, g_;
        if (admpc_rowqp_plan(cfg->N, 8, 1, s->num_cu, &r_, &st_, &lb_, &g_) != 0) { delete s; return fail(ADMPC_EINVAL, "horizon too long for the LDS-resident kernel"); }
    }
    s->cap = s->cap_lin = s->cap_dense = s->cap_row = 0; s->row_elem = 8; s->sched_cap = 0; s->d_tick = nullptr; s->tick_flip = 0; s->d_slot = nullptr; s->cap_fused = 0;
    s->d_GT = nullptr; s->d_bl = nullptr; s->d_status = nullptr; s->d_H = nullptr; s->d_aux = nullptr; s->d_ws = nullptr; s->d_pairs = nullptr; s->d_split = nullptr; s->d_dump = nullptr; s->d_mult = nullptr; s->cap_mult = 0; s->mult_elem = 0;
    {   // ADMPC_ROWQP_SPLIT=0 / 1: never / always run the row kernel in two phases (A/B tests); default: by batch size
        const char* e = getenv("ADMPC_ROWQP_SPLIT");
        s->split_mode = e && e[0] == '0' ? 0 : (e && e[0] == '1' ? 1 : -1);
        const char* c = getenv("ADMPC_ROWQP_CHUNK");
        s->row_chunk = c ? atoi(c) : 0;
    }
    {   // ADMPC_QP=riccati forces the stage-wise Riccati kernel (A/B tests); default: condensed kernel where instantiated
        const char* e = getenv("ADMPC_QP");
        s->use_dense = (cfg->N == 20) && !(e && strcmp(e, "riccati") == 0);
        s->dense_lds_bytes = DenseLds<20>::total * (int)sizeof(double);
        // ADMPC_N20=split keeps the four-kernel pipeline (linearise, condense, interior point, expand) for A/B runs and tests
#ifdef ADMPC_LEGACY_N20
        const char* m = getenv("ADMPC_N20");
        s->n20_fused = !(m && strcmp(m, "split") == 0);
#else
        s->n20_fused = 1;                        // the product library carries the fused kernel only (`make legacy` for the older pipeline)
#endif
        // default at N = 40 (the reference's shipped horizon: 6.2 M solves/s against kernel R's 3.4 M at B = 4096), 60 and 80 (2.7 / 2.0 M against
        // 1.7 / 1.2 M; scripts/cmp_seg_rowqp.sh); ADMPC_QP=riccati selects kernel R there
        // (nominal model only: with GP residuals in the dynamics the linearisation can have strongly unstable modes -- random regressors are
        // arbitrary dynamics -- and eliminating 20 stages at a time loses what the stage-wise Riccati recursion keeps: the N = 40 + GP census
        // family came out 3e-6 off in the inputs on the segmented kernel, 1e-8 on kernel R; with GPs the default stays kernel R, ADMPC_QP=seg selects kernel S)
        s->use_seg = admpc_seg_supports(cfg->N) && (cfg->n_gp == 0 ? !(e && strcmp(e, "riccati") == 0) : (e && strcmp(e, "seg") == 0));
    }
    hipError_t e = hipMalloc((void**)&s->d_cfg, sizeof(AdmpcConfig));
    if (e != hipSuccess) { delete s; return fail(ADMPC_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemcpy(s->d_cfg, cfg, sizeof(AdmpcConfig), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(s->d_cfg); delete s; return fail(ADMPC_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    s->d_sched = nullptr;
    s->qmask = 7;
    for (int c = 3; c < NX; ++c) if (cfg->W[c] != 0.0 || cfg->We[c] != 0.0) s->qmask = 127;
    // opt in to > 64 KB of dynamic LDS
    admpc_rowqp_prepare();
#ifdef ADMPC_LEGACY_N20
    (void)hipFuncSetAttribute((const void*)admpc_qp_dense_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    *out = s;
    return ADMPC_OK;
}

void admpc_destroy(AdmpcSolver* s)
{
    if (!s) return;
    DeviceGuard guard(s->device);
#ifdef ADMPC_PHASE_TIMERS
    {
        unsigned long long h[16] = {0};
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_ticks), sizeof h) == hipSuccess) {
            static const char* nm[16] = {"ipm staging", "ipm phase A", "ipm factorisation", "ipm phase C", "ipm substitutions", "ipm expand/step", "ipm hand-over", "ipm scheduler draw", "ipm unconstrained trial", "-",
                                         "cond staging", "cond H accumulate", "cond propagate", "cond store+bin", "-", "-"};
            unsigned long long tot = 0; for (int i = 0; i < 10; ++i) tot += h[i];
            unsigned long long totc = 0; for (int i = 10; i < 16; ++i) totc += h[i];
            for (int i = 0; i < 14; ++i) if (i < 9 || i >= 10) fprintf(stderr, "[admpc phase] %-18s %14llu ticks %5.1f %%\n", nm[i], h[i], 100.0 * (double)h[i] / (double)((i < 10 ? tot : totc) ? (i < 10 ? tot : totc) : 1));
        }
    }
#endif
    (void)hipFree(s->d_cfg);
    if (s->d_tick) (void)hipFree(s->d_tick);
    if (s->d_slot) (void)hipFree(s->d_slot);
    if (s->d_sched) (void)hipFree(s->d_sched);
    if (s->d_GT) (void)hipFree(s->d_GT);
    if (s->d_bl) (void)hipFree(s->d_bl);
    if (s->d_status) (void)hipFree(s->d_status);
    if (s->d_H) (void)hipFree(s->d_H);
    if (s->d_aux) (void)hipFree(s->d_aux);
    if (s->d_ws) (void)hipFree(s->d_ws);
    if (s->d_split) (void)hipFree(s->d_split);
    if (s->d_mult) (void)hipFree(s->d_mult);
    if (s->d_dump) (void)hipFree(s->d_dump);
    if (s->d_pairs) (void)hipFree(s->d_pairs);
    delete s;
}

// Kernel R addresses its arrays with 32-bit byte offsets from the (64-bit) array bases: a batch whose largest array would pass 4 GB
// is solved in consecutive chunks on the caller's stream (same results: instances are independent; the workspace is sized for one
// chunk).  ADMPC_ROWQP_CHUNK=n caps the chunk (tests).
static int rowqp_chunk(const AdmpcSolver* s, int N, int elem)
{
    const unsigned long long per = (unsigned long long)(N + 1) * 42ull * (unsigned long long)elem;
    unsigned long long m = ((1ull << 32) - 1ull) / per;
    if (m > 0x7fffffffull) m = 0x7fffffffull;
    if (s->row_chunk > 0 && (unsigned long long)s->row_chunk < m) m = (unsigned long long)s->row_chunk;
    return (int)m;
}

// ---- workspaces: grown on demand, each by the path that uses it.  Growing synchronises the device (nothing may still be using
//      the old block) and allocates; admpc_reserve does it up front for the handle's default path.
#define GROW(ptr, type, count) do { if (ptr) (void)hipFree(ptr); ptr = nullptr; HIPCHK(hipMalloc((void**)&(ptr), (size_t)(count) * sizeof(type))); } while (0)
static int ensure_status(AdmpcSolver* s, int B)
{
    if (B <= s->cap) return ADMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    s->cap = 0;
    GROW(s->d_status, int32_t, B);
    s->cap = B;
    return ADMPC_OK;
}
static int ensure_sched(AdmpcSolver* s, int lists_for)          // header (ticket counter, bucket counts) + optional scheduler lists
{
    if (s->d_sched && lists_for <= s->sched_cap) return ADMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    s->sched_cap = 0;
    GROW(s->d_sched, int, (size_t)SCHED_HDR + (size_t)SCHED_NB * (size_t)lists_for);
    s->sched_cap = lists_for;
    return ADMPC_OK;
}
static int ensure_lin(AdmpcSolver* s, int B)                    // kernel A's output
{
    int rc = ensure_sched(s, s->sched_cap); if (rc) return rc;
    if (B <= s->cap_lin) return ADMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    const size_t N = (size_t)s->cfg.N;
    s->cap_lin = 0;
    GROW(s->d_GT, double, (size_t)B * N * GTS);
    GROW(s->d_bl, double, (size_t)B * N * NX);
    s->cap_lin = B;
    return ADMPC_OK;
}
static int ensure_dense(AdmpcSolver* s, int B)                  // four-kernel N = 20 pipeline
{
    int rc = ensure_lin(s, B); if (rc) return rc;
    if (B <= s->cap_dense) return ADMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    s->cap_dense = 0;
    GROW(s->d_H, double, (size_t)B * DenseLds<20>::NTRI);
    GROW(s->d_aux, double, (size_t)B * 128);
    rc = ensure_sched(s, B); if (rc) return rc;
    s->cap_dense = B;
    return ADMPC_OK;
}
static int ensure_fused(AdmpcSolver* s, int B)                  // fused N = 20 step: one slot buffer per resident wave, the work-order lists
{                                                                // (segmented kernel: the work-order lists only)
    // slot buffers only where parking the linearisation beats recomputing it: with GP residuals in the model (see admpc_fused20.hip)
    // (ADMPC_F20_PARK=1 / 0 forces / forbids the slot buffers whatever the model: A/B runs)
    const char* pk = getenv("ADMPC_F20_PARK");
    const bool park = pk ? pk[0] == '1' : s->cfg.n_gp > 0;
    if (!s->d_slot && park && !s->use_seg) HIPCHK(hipMalloc((void**)&s->d_slot, admpc_fused20_slot_doubles(s->num_cu) * sizeof(double)));
    // segmented kernel: one packed Hessian per resident wave (its LDS buffer doubles as the factor's)
    if (!s->d_slot && s->use_seg) HIPCHK(hipMalloc((void**)&s->d_slot, admpc_seg_slot_doubles(s->num_cu) * sizeof(double)));
    if (B <= s->cap_fused) return ADMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    s->cap_fused = 0;
    GROW(s->d_tick, int, admpc_fused20_sched_ints(B));
    HIPCHK(hipMemset(s->d_tick, 0, admpc_fused20_sched_ints(B) * sizeof(int)));
    HIPCHK(hipDeviceSynchronize());                 // the memset runs on the null stream: the caller's (non-blocking) stream must not overtake it
    s->cap_fused = B;
    return ADMPC_OK;
}
static int ensure_row(AdmpcSolver* s, int B, int elem)          // kernel R; sized by the element width of the solve
{
    int rc = ensure_lin(s, B); if (rc) return rc;
    if (B <= s->cap_row && elem <= s->row_elem) return ADMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    const size_t N = (size_t)s->cfg.N;
    const int nb = B > s->cap_row ? B : s->cap_row;
    const int ne = (s->cap_row > 0 && s->row_elem > elem) ? s->row_elem : elem;
    s->cap_row = 0;
    if (s->d_ws) (void)hipFree(s->d_ws); s->d_ws = nullptr;
    if (s->d_dump) (void)hipFree(s->d_dump); s->d_dump = nullptr;
    HIPCHK(hipMalloc((void**)&s->d_ws, (size_t)nb * (N + 1) * 38 * (size_t)ne));          // RQ_RW = 38 values per record
    HIPCHK(hipMalloc((void**)&s->d_dump, (size_t)nb * (16 + 31 * N) * (size_t)ne));
    GROW(s->d_split, int32_t, (size_t)2 * nb + 1);
    s->cap_row = nb; s->row_elem = ne;
    return ADMPC_OK;
}

static int ensure_mult(AdmpcSolver* s, int B, int elem)         // SQP solves with a tolerance whose caller passes no multiplier arrays
{
    if (B <= s->cap_mult && elem <= s->mult_elem) return ADMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    const size_t N = (size_t)s->cfg.N;
    const int nb = B > s->cap_mult ? B : s->cap_mult;
    const int ne = elem > s->mult_elem ? elem : s->mult_elem;
    s->cap_mult = 0;
    if (s->d_mult) (void)hipFree(s->d_mult); s->d_mult = nullptr;
    HIPCHK(hipMalloc((void**)&s->d_mult, (size_t)nb * ((N + 1) * NX + 20 * N) * (size_t)ne));
    s->cap_mult = nb; s->mult_elem = ne;
    return ADMPC_OK;
}

int admpc_reserve(AdmpcSolver* s, int B)
{
    if (!s || B < 0) return fail(ADMPC_EINVAL, "admpc_reserve: bad argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    int rc = ensure_status(s, B); if (rc) return rc;
    const bool tol_on = s->cfg.sqp_iters > 1 && s->cfg.sqp_tol > 0.0;  // such solves run on the row kernel at every horizon (solve_impl)
    if (s->use_dense && s->n20_fused && !tol_on) return ensure_fused(s, B);      // no per-instance workspace
    if (s->use_seg && !tol_on) return ensure_fused(s, B);
    if (s->use_dense && !tol_on) return ensure_dense(s, B);
    const int chunk = rowqp_chunk(s, s->cfg.N, 8);
    rc = ensure_row(s, B < chunk ? B : chunk, 8); if (rc) return rc;
    return tol_on ? ensure_mult(s, B < chunk ? B : chunk, 8) : ADMPC_OK;
}

// Two phases for the row kernel (admpc_rowqp.hip)?  Only with the unconstrained trial on; by default when the batch is more than one
// round of waves (up to one round every instance starts at t = 0 anyway: -4 .. +4 % measured; at 1.2 rounds the split already wins 18 %).
static int32_t* rowqp_split(const AdmpcSolver* s, int B, int rows, int grid)
{
    if (s->cfg.ipm_try_unconstrained == 0.0 || s->split_mode == 0) return nullptr;
    const int nquads = (B + rows - 1) / rows;
    return (s->split_mode == 1 || nquads > grid) ? s->d_split : nullptr;
}

}  // extern "C"

// linearisation (kernel A) + row-mapped Riccati interior point (kernel R) for a batch of any size, T = double or float
template <class T>
static int solve_rows(AdmpcSolver* s, int B, const T* x0, const T* yref, const T* yref_e, const T* p, T* xbar, T* ubar,
                      T* cost, int32_t* stat, int32_t* iters, T* pi, T* ineq, hipStream_t st, int routed = 0)
{
    const int N = s->cfg.N, elem = (int)sizeof(T);
    const int chunk = rowqp_chunk(s, N, elem);
    const int nsqp = s->cfg.sqp_iters > 0 ? s->cfg.sqp_iters : 1;
    const bool tol_on = nsqp > 1 && s->cfg.sqp_tol > 0.0;      // acados' residual test in front of every QP but the first (admpc_nlp_res_kernel)
    { int rc = ensure_row(s, B < chunk ? B : chunk, elem); if (rc) return rc; }
    if (tol_on && !pi) { int rc = ensure_mult(s, B < chunk ? B : chunk, elem); if (rc) return rc; }
    for (long off = 0; off < (long)B; off += chunk) {
        const int nb = (long)B - off < (long)chunk ? (int)((long)B - off) : chunk;
        int rows, stride, ldsb, gridR;
        if (admpc_rowqp_plan(N, elem, nb, s->num_cu, &rows, &stride, &ldsb, &gridR) != 0) return fail(ADMPC_EINVAL, "horizon too long for the LDS-resident kernel");
        const long totalA = (long)nb * N * 3;
        int gridA = (int)((totalA + LIN_TASKS - 1) / LIN_TASKS);
        if (gridA > s->num_cu * 64) gridA = s->num_cu * 64;
        const T* cx0 = x0 + off * NX; const T* cyr = yref + off * N * NY; const T* cye = yref_e + off * NX; const T* cp = p + off;
        T* cxb = xbar + off * (N + 1) * NX; T* cub = ubar + off * N * NU;
        T* cco = cost ? cost + off : nullptr; int32_t* cst = stat + off; int32_t* cit = iters ? iters + off : nullptr;
        T* cpi = pi ? pi + off * (N + 1) * NX : nullptr; T* ciq = ineq ? ineq + off * N * 20 : nullptr;
        if (tol_on && !pi) { cpi = (T*)s->d_mult; ciq = cpi + (size_t)nb * (N + 1) * NX; }      // the chunk's multipliers live between its passes only
        for (int sq = 0; sq < nsqp; ++sq) {
            const int first = (sq == 0 && !routed) ? 1 : 0;      // routed: the status array says which instances are this handle's (0) from the start
            hipLaunchKernelGGL(admpc_linearize_kernel<T>, dim3(gridA), dim3(LIN_BLOCK), 0, st, s->d_cfg, nb, (const T*)cxb, (const T*)cub, cp,
                               first ? (const int32_t*)nullptr : (const int32_t*)cst, (T*)s->d_GT, (T*)s->d_bl, s->d_sched);
            if (tol_on && sq > 0)
                hipLaunchKernelGGL(admpc_nlp_res_kernel<T>, dim3(nb < s->num_cu * 32 ? nb : s->num_cu * 32), dim3(WAVE), 0, st, s->d_cfg, nb, cx0, cyr, cye,
                                   (const T*)cxb, (const T*)cub, (const T*)s->d_GT, (const T*)s->d_bl, (const T*)cpi, (const T*)ciq, cst, (T*)nullptr);
            if constexpr (sizeof(T) == 8)
                admpc_rowqp_launch_f64(gridR, ldsb, st, s->d_cfg, nb, rows, stride, cx0, cyr, cye, (const double*)s->d_GT, (const double*)s->d_bl,
                                       cxb, cub, cco, cst, cit, cpi, ciq, s->d_ws, first, s->d_sched, rowqp_split(s, nb, rows, gridR), s->d_dump);
            else
                admpc_rowqp_launch_f32(gridR, ldsb, st, s->d_cfg, nb, rows, stride, cx0, cyr, cye, (const float*)s->d_GT, (const float*)s->d_bl,
                                       cxb, cub, cco, cst, cit, cpi, ciq, (float*)s->d_ws, first, s->d_sched, rowqp_split(s, nb, rows, gridR), (float*)s->d_dump);
        }
    }
    return ADMPC_OK;
}

extern "C" {

int admpc_solve_batch_ex(AdmpcSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* p,
                         double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, double* pi, double* ineq, void* stream);

int admpc_solve_batch(AdmpcSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* p,
                      double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream)
{
    return admpc_solve_batch_ex(s, B, x0, yref, yref_e, p, xbar, ubar, cost, status, iters, nullptr, nullptr, stream);
}

/* admpc_solve_batch plus the multipliers of the returned iterate (acados store_iterate contents).  With pi / ineq given the step
 * runs on the row kernel at every horizon (the condensed N = 20 pipeline eliminates the states and carries no multipliers of the
 * dynamics).  routed (admpc_solve_batch_routed): `status` arrives filled -- 0 for the instances this handle is to solve, non-zero for
 * the others, which every kernel then leaves alone (the mechanism that skips failed / converged instances in later SQP passes). */
static int solve_impl(AdmpcSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* p,
                      double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters,
                      double* pi, double* ineq, void* stream, int routed)
{
    const bool snap = pi != nullptr || ineq != nullptr;
    if (snap && !(pi && ineq)) return fail(ADMPC_EINVAL, "pi and ineq must be given together");
    if (!s) return fail(ADMPC_EINVAL, "null solver");
    if (B < 0) return fail(ADMPC_EINVAL, "negative batch");
    if (B == 0) return ADMPC_OK;
    if (!x0 || !yref || !yref_e || !p || !xbar || !ubar) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    const int N = s->cfg.N;
    const int nsqp = s->cfg.sqp_iters > 0 ? s->cfg.sqp_iters : 1;
    // the condensed N = 20 kernels carry no multipliers: solves that return them, and SQP solves with a tolerance (whose stopping test
    // needs them between the passes), run on the row kernel at every horizon
    const bool dense = s->use_dense && !snap && !(nsqp > 1 && s->cfg.sqp_tol > 0.0), fused = dense && s->n20_fused;
    {   // workspaces of the path this call takes (no-ops once sized: admpc_reserve up front keeps the default path allocation-free)
        int rc = ensure_status(s, B); if (rc) return rc;
        if (fused) { rc = ensure_fused(s, B); if (rc) return rc; }
        else if (dense) { rc = ensure_dense(s, B); if (rc) return rc; }
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t* stat = status ? status : s->d_status;
    if (s->use_seg && !snap && !(nsqp > 1 && s->cfg.sqp_tol > 0.0)) {
        // N = 40 / 60 / 80: S cooperating waves per instance, each condensing 20 stages; no workspace, no kernel boundary (admpc_seg.hip)
        int rc = ensure_fused(s, B); if (rc) return rc;
        for (int sq = 0; sq < nsqp; ++sq)
            admpc_seg_launch(N, s->num_cu, st, s->d_cfg, B, s->qmask, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, (sq == 0 && !routed) ? 1 : 0, s->d_tick, s->cap_fused, (s->tick_flip ^= 1), s->d_slot);
        HIPCHK(hipGetLastError());
        return ADMPC_OK;
    }
    if (!dense) {      // every horizon but N = 20, and every solve that asks for multipliers: kernel A + kernel R, in chunks if need be
        int rc = solve_rows<double>(s, B, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, pi, ineq, st, routed); if (rc) return rc;
        if (nsqp > 1 && s->cfg.sqp_tol > 0.0)
            hipLaunchKernelGGL(admpc_sqp_finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, stat);
        HIPCHK(hipGetLastError());
        return ADMPC_OK;
    }
    const long totalA = (long)B * N * 3;
    int gridA = (int)((totalA + LIN_TASKS - 1) / LIN_TASKS);
    if (gridA > s->num_cu * 64) gridA = s->num_cu * 64;
    for (int sq = 0; sq < nsqp; ++sq) {
        const int first = (sq == 0 && !routed) ? 1 : 0;
        if (fused) {
            // shooting, condensing, interior point and expansion of an instance in one persistent kernel: no workspace, no kernel boundary
            admpc_fused20_launch(s->num_cu, st, s->d_cfg, B, s->qmask, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, s->d_tick, s->cap_fused, (s->tick_flip ^= 1), s->d_slot);
            continue;
        }
#ifdef ADMPC_LEGACY_N20      // the four-kernel pipeline of rounds 1-2 (`make legacy`, ADMPC_N20=split)
        hipLaunchKernelGGL(admpc_linearize_kernel<double>, dim3(gridA), dim3(LIN_BLOCK), 0, st, s->d_cfg, B, xbar, ubar, p,
                           first ? (const int32_t*)nullptr : (const int32_t*)stat, s->d_GT, s->d_bl, s->d_sched);
        {
            constexpr int cond_lds = (DenseLds<20>::NTRI + (DenseLds<20>::NTRI & 1) + 20 * GTS + DenseLds<20>::BLS + DenseLds<20>::DQS + NX * 64) * (int)sizeof(double);
            int gridC = s->num_cu * 8; if (gridC > B) gridC = B;
            if (s->qmask == 7)
                hipLaunchKernelGGL((admpc_condense_kernel<20, 7>), dim3(gridC), dim3(WAVE), cond_lds, st, s->d_cfg, B, x0, yref, yref_e,
                                   (const double*)s->d_GT, (const double*)s->d_bl, (const double*)xbar, (const double*)ubar,
                                   (const int32_t*)stat, first, s->d_H, s->d_aux, s->d_sched, s->sched_cap);
            else
                hipLaunchKernelGGL((admpc_condense_kernel<20, 127>), dim3(gridC), dim3(WAVE), cond_lds, st, s->d_cfg, B, x0, yref, yref_e,
                                   (const double*)s->d_GT, (const double*)s->d_bl, (const double*)xbar, (const double*)ubar,
                                   (const int32_t*)stat, first, s->d_H, s->d_aux, s->d_sched, s->sched_cap);
            int gridD = s->num_cu * ((160 * 1024) / s->dense_lds_bytes < 8 ? (160 * 1024) / s->dense_lds_bytes : 8);   // two waves per SIMD
            if (gridD > B) gridD = B;
            hipLaunchKernelGGL((admpc_qp_dense_kernel<20>), dim3(gridD), dim3(WAVE), s->dense_lds_bytes, st, s->d_cfg, B,
                               (const double*)xbar, (const double*)ubar, cost, stat, iters,
                               (const double*)s->d_H, s->d_aux, s->d_sched, s->sched_cap);
            constexpr int exp_lds = DenseLds<20>::expand_total * (int)sizeof(double);
            int gridE = s->num_cu * 16; if (gridE > B) gridE = B;
            hipLaunchKernelGGL((admpc_expand_kernel<20>), dim3(gridE), dim3(WAVE), exp_lds, st, s->d_cfg, B, x0, yref, yref_e,
                               (const double*)s->d_GT, (const double*)s->d_bl, xbar, ubar, cost, stat, (const double*)s->d_aux);
        }
#else
        (void)gridA;
#endif
    }
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_solve_batch_ex(AdmpcSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* p,
                         double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters,
                         double* pi, double* ineq, void* stream)
{
    return solve_impl(s, B, x0, yref, yref_e, p, xbar, ubar, cost, status, iters, pi, ineq, stream, 0);
}

// ---- clustered GP ensembles (reference: one solver per cluster, chosen per solve by GPEnsemble.select_gp)
namespace {
#pragma clang fp contract(off)
// nearest centroid of z = [x; u][feats] (Euclidean distance as numpy / the reference computes it, ties -> lowest index): gp.py:738-770
__global__ void admpc_select_cluster_kernel(int B, int d, int f0, int f1, int f2, const double* __restrict__ xs, const double* __restrict__ us,
                                            int K, const double* __restrict__ cent, int32_t* __restrict__ route)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int f[3] = { f0, f1, f2 };
    double z[3];
    for (int j = 0; j < d; ++j) z[j] = f[j] < NX ? xs[(size_t)b * NX + f[j]] : us[(size_t)b * NU + f[j] - NX];
    double best = INFINITY; int bi = 0;
    for (int c = 0; c < K; ++c) {
        double acc = 0.0;
        for (int j = 0; j < d; ++j) { const double e = z[j] - cent[c * d + j]; acc = acc + e * e; }
        const double dist = __dsqrt_rn(acc);
        if (dist < best) { best = dist; bi = c; }
    }
    route[b] = bi;
}
// status array of one cluster's solve: 0 = this handle's instance, ADMPC_STATUS_SKIP = somebody else's
#define ADMPC_STATUS_SKIP (-2)
__global__ void admpc_route_fill_kernel(int B, const int32_t* __restrict__ route, int c, int32_t* __restrict__ tmp)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) tmp[b] = route[b] == c ? 0 : ADMPC_STATUS_SKIP;
}
__global__ void admpc_route_merge_kernel(int B, const int32_t* __restrict__ route, int c, const int32_t* __restrict__ tmp, int32_t* __restrict__ status)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && route[b] == c) status[b] = tmp[b];
}
// an instance routed to no cluster at all is reported as failed instead of being passed over in silence
__global__ void admpc_route_invalid_kernel(int B, const int32_t* __restrict__ route, int K, int32_t* __restrict__ status, double* __restrict__ cost, int32_t* __restrict__ iters)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (route[b] < 0 || route[b] >= K)) { if (status) status[b] = ADMPC_STATUS_QP_FAILURE; if (cost) cost[b] = INFINITY; if (iters) iters[b] = 0; }
}
}

int admpc_select_cluster_batch(int device, int B, int n_feat, const int32_t* feats, const double* x_sel, const double* u_sel,
                               int K, const double* centroids, int32_t* route, void* stream)
{
    if (B < 0 || n_feat < 1 || n_feat > 3 || !feats || K < 1) return fail(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!x_sel || !u_sel || !centroids || !route) return fail(ADMPC_EINVAL, "null array argument");
    for (int j = 0; j < n_feat; ++j) if (feats[j] < 0 || feats[j] >= NX + NU) return fail(ADMPC_EINVAL, "feature index outside [x; u]");
    DeviceGuard guard(device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(admpc_select_cluster_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, n_feat, feats[0], n_feat > 1 ? feats[1] : 0,
                       n_feat > 2 ? feats[2] : 0, x_sel, u_sel, K, centroids, route);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_solve_batch_routed(AdmpcSolver* const* solvers, int K, int B, const int32_t* route,
                             const double* x0, const double* yref, const double* yref_e, const double* p,
                             double* xbar, double* ubar, double* cost, int32_t* status, int32_t* iters, void* stream)
{
    if (!solvers || K < 1) return fail(ADMPC_EINVAL, "bad argument");
    if (B < 0) return fail(ADMPC_EINVAL, "negative batch");
    if (B == 0) return ADMPC_OK;
    if (!route || !x0 || !yref || !yref_e || !p || !xbar || !ubar) return fail(ADMPC_EINVAL, "null array argument");
    for (int c = 0; c < K; ++c) {
        if (!solvers[c]) return fail(ADMPC_EINVAL, "null solver");
        if (solvers[c]->device != solvers[0]->device || solvers[c]->cfg.N != solvers[0]->cfg.N) return fail(ADMPC_EINVAL, "the cluster solvers must share device and horizon");
    }
    DeviceGuard guard(solvers[0]->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((B + 255) / 256), blk(256);
    hipLaunchKernelGGL(admpc_route_invalid_kernel, g, blk, 0, st, B, route, K, status, cost, iters);
    for (int c = 0; c < K; ++c) {
        AdmpcSolver* s = solvers[c];
        int rc = ensure_status(s, B); if (rc) return rc;
        hipLaunchKernelGGL(admpc_route_fill_kernel, g, blk, 0, st, B, route, c, s->d_status);
        rc = solve_impl(s, B, x0, yref, yref_e, p, xbar, ubar, cost, s->d_status, iters, nullptr, nullptr, stream, 1); if (rc) return rc;
        if (status) hipLaunchKernelGGL(admpc_route_merge_kernel, g, blk, 0, st, B, route, c, (const int32_t*)s->d_status, status);
    }
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

/* fp32 storage and arithmetic (BASELINE configs[4]): same arguments as admpc_solve_batch with float arrays. */
int admpc_solve_batch_f32(AdmpcSolver* s, int B, const float* x0, const float* yref, const float* yref_e, const float* p,
                          float* xbar, float* ubar, float* cost, int32_t* status, int32_t* iters, void* stream)
{
    if (!s) return fail(ADMPC_EINVAL, "null solver");
    if (B < 0) return fail(ADMPC_EINVAL, "negative batch");
    if (B == 0) return ADMPC_OK;
    if (!x0 || !yref || !yref_e || !p || !xbar || !ubar) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    { int rc = ensure_status(s, B); if (rc) return rc; }
    hipStream_t st = (hipStream_t)stream;
    int32_t* stat = status ? status : s->d_status;
    const int nsqp = s->cfg.sqp_iters > 0 ? s->cfg.sqp_iters : 1;
    { int rc = solve_rows<float>(s, B, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, (float*)nullptr, (float*)nullptr, st); if (rc) return rc; }
    if (nsqp > 1 && s->cfg.sqp_tol > 0.0)
        hipLaunchKernelGGL(admpc_sqp_finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, stat);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_nlp_residuals_batch(AdmpcSolver* s, int B, const double* x0, const double* yref, const double* yref_e, const double* p,
                              const double* xbar, const double* ubar, const double* pi, const double* ineq, double* res, void* stream)
{
    if (!s) return fail(ADMPC_EINVAL, "null solver");
    if (B < 0) return fail(ADMPC_EINVAL, "negative batch");
    if (B == 0) return ADMPC_OK;
    if (!x0 || !yref || !yref_e || !p || !xbar || !ubar || !pi || !ineq || !res) return fail(ADMPC_EINVAL, "null array argument");
    const int N = s->cfg.N;
    if ((double)B * N * GTS * 8.0 > 4.0e9) return fail(ADMPC_EINVAL, "batch too large for one linearisation: split it");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    { int rc = ensure_status(s, B); if (rc) return rc; rc = ensure_lin(s, B); if (rc) return rc; }
    hipStream_t st = (hipStream_t)stream;
    const long totalA = (long)B * N * 3;
    int gridA = (int)((totalA + LIN_TASKS - 1) / LIN_TASKS);
    if (gridA > s->num_cu * 64) gridA = s->num_cu * 64;
    HIPCHK(hipMemsetAsync(s->d_status, 0, (size_t)B * sizeof(int32_t), st));
    hipLaunchKernelGGL(admpc_linearize_kernel<double>, dim3(gridA), dim3(LIN_BLOCK), 0, st, s->d_cfg, B, xbar, ubar, p,
                       (const int32_t*)nullptr, s->d_GT, s->d_bl, s->d_sched);
    hipLaunchKernelGGL(admpc_nlp_res_kernel<double>, dim3(B < s->num_cu * 32 ? B : s->num_cu * 32), dim3(WAVE), 0, st, s->d_cfg, B, x0, yref, yref_e,
                       xbar, ubar, (const double*)s->d_GT, (const double*)s->d_bl, pi, ineq, s->d_status, res);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_shoot_batch(AdmpcSolver* s, int B, const double* xbar, const double* ubar, const double* p,
                      double* phi, double* A, double* Bm, void* stream)
{
    if (!s || B < 0) return fail(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!xbar || !ubar || !p || !phi || !A || !Bm) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    long total = (long)B * s->cfg.N * 3;
    int grid = (int)((total + LIN_TASKS - 1) / LIN_TASKS); if (grid > s->num_cu * 32) grid = s->num_cu * 32;
    hipLaunchKernelGGL(admpc_shoot_kernel, dim3(grid), dim3(WAVE), 0, (hipStream_t)stream, s->d_cfg, B, xbar, ubar, p, phi, A, Bm);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_argmin(AdmpcSolver* s, const double* cost, int B, int64_t index_offset, double* val, int64_t* idx, void* stream)
{
    if (!s || !cost || !val || !idx || B <= 0) return fail(ADMPC_EINVAL, "bad argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(admpc_argmin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cost, B, index_offset, val, idx);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_argmin_pairs(AdmpcSolver* s, const double* pairs, int W, double* val, int64_t* idx, void* stream)
{
    if (!s || W <= 0) return fail(ADMPC_EINVAL, "bad argument");
    if (!pairs || !val || !idx) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(admpc_argmin_pairs_kernel, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, pairs, W, val, idx);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

/* Host-side twin of admpc_argmin_pairs for records that were gathered into HOST memory (gloo / MPI hosts; the CPU tests of the
 * N > 1 path): same rules, same code (argmin_rule.h).  No device, no solver handle. */
int admpc_argmin_pairs_host(const double* pairs, int W, double* val, int64_t* idx)
{
    if (W <= 0) return fail(ADMPC_EINVAL, "bad argument");
    if (!pairs || !val || !idx) return fail(ADMPC_EINVAL, "null array argument");
    ArgminBest b = argmin_identity();
    for (int i = 0; i < W; ++i) {
        int64_t ix; memcpy(&ix, pairs + 2 * i + 1, sizeof ix);
        argmin_fold(b, argmin_cost(pairs[2 * i]), ix);
    }
    *val = b.v; *idx = argmin_final_index(b);
    return ADMPC_OK;
}

// RCCL is resolved at the first call (dlopen): the library loads and every other entry point works on a host without it.
namespace {
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_count_t)(void*, int*);
struct RcclApi { nccl_allgather_t all_gather; nccl_count_t comm_count; };
RcclApi g_rccl = { nullptr, nullptr };
std::once_flag g_rccl_once;
// The caller's ncclComm_t belongs to the RCCL copy that created it -- in a torch process that is the librccl torch bundles
// (SONAME librccl.so.1), not necessarily the one under /opt/rocm.  Look the symbols up in what the process has ALREADY loaded
// first (global scope, then the two sonames without loading anything); map a fresh copy only when none is there.
void rccl_lookup() {
    void* ag = dlsym(RTLD_DEFAULT, "ncclAllGather");
    void* cc = dlsym(RTLD_DEFAULT, "ncclCommCount");
    if (!(ag && cc)) {
        static const char* names[2] = { "librccl.so.1", "librccl.so" };
        for (int pass = 0; pass < 2 && !(ag && cc); ++pass)           // pass 0: RTLD_NOLOAD (already mapped copies only)
            for (int k = 0; k < 2 && !(ag && cc); ++k) {
                void* h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (!h) continue;
                ag = dlsym(h, "ncclAllGather"); cc = dlsym(h, "ncclCommCount");
            }
    }
    if (ag && cc) { g_rccl.all_gather = (nccl_allgather_t)ag; g_rccl.comm_count = (nccl_count_t)cc; }
}
}

int admpc_argmin_global(AdmpcSolver* s, const double* cost, int B, int64_t index_offset, void* nccl_comm,
                        double* val, int64_t* idx, void* stream)
{
    if (!s || !cost || !val || !idx || B <= 0 || !nccl_comm) return fail(ADMPC_EINVAL, "bad argument");
    std::call_once(g_rccl_once, rccl_lookup);
    if (!g_rccl.all_gather || !g_rccl.comm_count) return fail(ADMPC_ENODEV, "librccl.so (ncclAllGather, ncclCommCount) not available");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    int nranks = 0;
    if (g_rccl.comm_count(nccl_comm, &nranks) != 0 || nranks < 1 || nranks > 256) return fail(ADMPC_EINVAL, "ncclCommCount failed or more than 256 ranks");
    if (!s->d_pairs) HIPCHK(hipMalloc((void**)&s->d_pairs, (size_t)(1 + 256) * 2 * sizeof(double)));
    hipStream_t st = (hipStream_t)stream;
    double* mine = s->d_pairs;                       // (cost, index bits)
    double* all = s->d_pairs + 2;
    hipLaunchKernelGGL(admpc_argmin_kernel, dim3(1), dim3(256), 0, st, cost, B, index_offset, mine, (int64_t*)(mine + 1));
    if (g_rccl.all_gather(mine, all, 2, 8 /* ncclFloat64 */, nccl_comm, st) != 0) return fail(ADMPC_EHIP, "ncclAllGather failed");
    hipLaunchKernelGGL(admpc_argmin_pairs_kernel, dim3(1), dim3(WAVE), 0, st, (const double*)all, nranks, val, idx);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_shift_batch(AdmpcSolver* s, int B, double* xbar, double* ubar, const double* p, int rollout, void* stream)
{
    if (!s || B < 0) return fail(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!xbar || !ubar || (rollout && !p)) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    const long total = (long)B * 3;
    long grid = (total + LIN_TASKS - 1) / LIN_TASKS;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(admpc_shift_kernel, dim3((unsigned)grid), dim3(WAVE), 0, (hipStream_t)stream, s->d_cfg, B, xbar, ubar, p, rollout ? 1 : 0);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_epilogue_batch(AdmpcSolver* s, int B, const double* xopt, const double* uopt, const double* xref_xy,
                         float* ack, int32_t* valid, void* stream)
{
    if (!s || B < 0) return fail(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!xopt || !uopt || !xref_xy || !ack || !valid) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(admpc_epilogue_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->cfg.N, B, xopt, uopt, xref_xy, ack, valid);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_actuation_batch(AdmpcSolver* s, int B, const double* xopt, const double* uopt, const double* xref_xy, const int32_t* status,
                          const double* steer_meas, int32_t* safe_count, int threshold, double* cost_io,
                          float* ack, int32_t* mode, int32_t* valid, void* stream)
{
    if (!s || B < 0 || threshold < 0) return fail(ADMPC_EINVAL, "bad argument");
    if (B == 0) return ADMPC_OK;
    if (!xopt || !uopt || !xref_xy || !status || !steer_meas || !safe_count || !ack || !mode || !valid) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(s->device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(admpc_actuation_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->cfg.N, B, xopt, uopt, xref_xy, status,
                       steer_meas, safe_count, threshold, s->cfg.lbu[1], s->cfg.ubu[1], s->cfg.lbx_delta, s->cfg.ubx_delta, cost_io, ack, mode, valid);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_resample_vel_batch(int device, int B, int H, int ld, const double* vx, const double* vy, double acc_max, double dt,
                             double* vel_ref, void* stream)
{
    if (B < 0 || H < 1 || ld < H) return fail(ADMPC_EINVAL, "admpc_resample_vel_batch: need H >= 1, ld >= H");
    if (B == 0) return ADMPC_OK;
    if (!vx || !vy || !vel_ref) return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    hipLaunchKernelGGL(admpc_resample_vel_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, H, ld, vx, vy, acc_max, dt, vel_ref);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

int admpc_waypoints_batch(int device, int M, int H, double dt, int B,
                          const double* vel, const double* x, const double* y, const double* psi, const double* psi_unwrapped,
                          const double* cdist, const double* curv,
                          const double* X_init, const double* Y_init, const double* psi_init,
                          double* out_ref, double* out_err, int32_t* out_stop, void* stream)
{
    if (M < 2 || H < 3 || H > WAVE || B < 0 || !(dt > 0)) return fail(ADMPC_EINVAL, "admpc_waypoints_batch: need M >= 2, 3 <= H <= 64, dt > 0");
    if (B == 0) return ADMPC_OK;
    if (!vel || !x || !y || !psi || !psi_unwrapped || !cdist || !curv || !X_init || !Y_init || !psi_init || !out_ref || !out_err || !out_stop)
        return fail(ADMPC_EINVAL, "null array argument");
    DeviceGuard guard(device);
    if (!guard.ok()) return fail(ADMPC_EHIP, "hipSetDevice failed");
    int grid = B < 4096 ? B : 4096;
    hipLaunchKernelGGL(admpc_waypoints_kernel, dim3(grid), dim3(WAVE), 0, (hipStream_t)stream, M, H, dt, B, vel, x, y, psi, psi_unwrapped, cdist, curv,
                       X_init, Y_init, psi_init, out_ref, out_err, out_stop);
    HIPCHK(hipGetLastError());
    return ADMPC_OK;
}

}  // extern "C"
