// admpc_seg.hip -- the fp64 SQP-RTI step for horizons N = 20 S (S = 2: the reference's shipped N = 40, launch/gp_ad_mpc.launch:6-7,
// acados_solver_sim_car.h:66; S = 3, 4: N = 60, 80) as ONE persistent kernel for gfx950: S cooperating wavefronts per MPC instance.
//
// The horizon is cut into S segments of 20 stages.  Wave s owns segment s from its inputs to its outputs with the machinery of the fused
// N = 20 kernel (admpc_fused20.hip): ERK4 + forward sensitivities of its stages (H0/H1, ad_3d_optimizer.py:280-310), condensing of its
// stages onto its 40 inputs (H2-H4, acados FULL_CONDENSING_HPIPM acados_solver_sim_car.c:145 -- per segment), the dense 40 x 40 LDL' of
// its Newton matrix with row i in the registers of lane i (H5, HPIPM :688-692), expansion and full step of its states (H6, :647-648).
// What couples the segments is the 7-vector at each cut, z_s = dx at the first stage of segment s (z_0 = x0 - xbar_0 is data):
//     z_{s+1} = Bbar_s U_s + Abar_s z_s + c_s          (multiplier nu_{s+1})
// The states of segment s are functions of (z_s, U_s): its condensing carries 7 more columns (Phi_k = d x_k / d z_s, lanes 40..46), its
// cost has the blocks Huu (40 x 40), Hzu (7 x 40), Hzz (7 x 7).  Per interior-point iteration every wave factorises its OWN Newton matrix
// M_s = Huu + barrier terms with the border rows C_s = [Qzu_s ; Bbar_s] riding along in lanes 40.. (a right-looking LDL' updates whatever
// rows the lanes hold: L_b = C L^-T D^-1 costs no instruction), the Schur blocks C M^-1 C' = L_b D L_b' come out of ten
// v_mfma_f64_16x16x4_f64, the border lanes of the forward substitution deliver the reduced right-hand sides, and wave 0 couples the
// segments by a backward / forward recursion over the S - 1 cuts in 7 x 7 blocks (Gaussian elimination with partial pivoting on
// Lambda = I + Pbb Pi); every wave then back-substitutes its own inputs.  All waves work at once; seven workgroup barriers per iteration.
// The Newton steps are the ones of the stage-wise Riccati oracle (oracle/admpc_oracle.c: ipm_solve) in another elimination order; the
// executable statement of the algebra is tests/seg_spec.py, checked against the oracle on the CPU (tests/test_seg_spec.py).
//
// Nothing but the algorithmic inputs and outputs crosses HBM (no workspace): the linearisation is recomputed in front of the expansion.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <math.h>
#include "../../include/admpc.h"

#define NX ADMPC_NX
#define NU ADMPC_NU
#define NY ADMPC_NY
#define WAVE 64
#define IPM_FLOOR 1e-40
#define GTS 42           // values per stage of the packed linearisation (see kernel A in admpc_kernels.hip)

#ifndef SEG_CPREF
#define SEG_CPREF 0
#endif
extern "C" size_t admpc_fused20_state_ints(int cap);      // admpc_fused20.hip: ints of one scheduler state
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

#include "model_dev.h"
#include "dense40.h"
#include "cond_common.h"
#include "work_order.h"
#include "seg_cut.h"          // bordered factorisation, Schur blocks, the operators of a single cut: shared with the quadrotor's segmented kernel

#define LDG(p) (*(p))
#define STG(p, v) (*(p) = (v))

// a wave-uniform double that the compiler cannot see to be uniform (an LDS read every lane does alike): into scalar registers -- the
// kernel lives at the 256-VGPR boundary of two waves per SIMD, and a scalar costs a vector register pair otherwise
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// The wave's H (S = 4: and Hb behind it) from its slot in global memory (L2) straight into LDS: LDS-DMA (global_load_lds_dwordx4: destination =
// wave-uniform base + lane * 16, no staging registers), issued as soon as the factor that occupies the buffer is dead -- behind the last back
// substitution of an iteration -- so that the round trip hides under the step-length computations.  The caller waits vmcnt(0) before the first read.
template <int CNT>
__device__ __forceinline__ void slot_fetch(double* lds_dst, const double* gsrc, const int lane) {
    constexpr int BYTES = CNT * 8, FULL = BYTES / 1024, REM = (BYTES % 1024) / 16;
    static_assert(BYTES % 16 == 0, "slot_fetch copies 16 bytes per lane");
    // 1 KiB pieces; four per base address (the instruction's 12-bit offset moves both the source and the LDS destination).  The base is laundered
    // where the copy is issued: hipcc otherwise hoists one 64-bit address per piece out of the interior-point loop and spills them.
    const char* g = reinterpret_cast<const char*>(gsrc) + lane * 16;
    asm volatile("" : "+v"(g));
    char* l = reinterpret_cast<char*>(lds_dst);
    static_for<0, FULL + (REM > 0 ? 1 : 0)>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value, grp = p / 4, off = (p % 4) * 1024;
        if (p < FULL || lane < REM)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + grp * 4096), (__attribute__((address_space(3))) void*)(l + grp * 4096), 16, off, 0);
    });
}

// Cross-WAVE exchange through LDS: inline-assembly LDS stores are invisible to the compiler's wait-count tracking, so the barrier
// waits for everything this wave has in flight first.
#define XSYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); } while (0)

// ---- bring-up dumps (build with -DSEG_DEBUG: `make variant`-style, scripts/seg_debug.py): instance SEG_DEBUG_INST leaves its LDS regions and
//      a few per-lane values in a global buffer at two points (P1: behind the condensing, P2: behind the unconstrained trial)
#ifdef SEG_DEBUG
#define SEG_DBG_W 16384
__device__ double g_seg_dbg[4 * 2 * SEG_DBG_W];
__device__ int g_seg_dbg_inst = 0;
__device__ __forceinline__ void seg_dbg_copy(double* dst, const double* src, int cnt, int lane) { for (int i = lane; i < cnt; i += 64) dst[i] = src[i]; }
#endif

// ---- optional per-phase wave-time accounting (-DADMPC_PHASE_TIMERS: `make timers`), s_memtime ticks (100 MHz) summed over all waves:
//      0 ticket, 1 phase A, 2 phase C, 3 trial, 4 iteration top (residuals, exchange), 5 factorisation + Schur blocks, 6 forward substitution,
//      7 wait for / run the interface recursion, 8 back substitution, 9 expand / step, 10 cut states, 11 phase A again, 12 phase E,
//      13 (S = 2) wait for the Schur blocks + the cut's solution operators (wave 0)
#ifdef ADMPC_PHASE_TIMERS
__device__ unsigned long long g_seg_trace[4 * 16384];     // per instance (first 16384): start, end (s_memrealtime, 100 MHz), workgroup, interior-point start
__device__ __forceinline__ unsigned long long seg_real() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ unsigned long long g_seg_ticks[16];
__device__ __forceinline__ unsigned long long seg_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
#define SEG_DECL() unsigned long long ph_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_last = seg_now()
#define SEG_STAMP(k) do { const unsigned long long t_ = seg_now(); ph_acc[k] += t_ - ph_last; ph_last = t_; } while (0)
#define SEG_FLUSH() do { if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 14; ++q_) atomicAdd(&g_seg_ticks[q_], ph_acc[q_]); } } while (0)
#elif defined(SEG_MARKS)      // listing with phase markers (hipcc -S -DSEG_MARKS): the code after stamp k belongs to the next phase
#define SEG_DECL() do { } while (0)
#define SEG_STAMP(k) asm volatile("; MARK_after" #k)
#define SEG_FLUSH() do { } while (0)
#else
#define SEG_DECL() do { } while (0)
#define SEG_STAMP(k) do { } while (0)
#define SEG_FLUSH() do { } while (0)
#endif

// dq[k][c] = xbar[k][c] - (k < 20 ? yref[k][c] : yre[c]), k = 0..20, for the 21 stage rows a segment touches
__device__ __forceinline__ void stage_dq_seg(double* __restrict__ dq, const double* __restrict__ xb, const double* __restrict__ yr,
                                             const double* __restrict__ yre, const int lane) {
    constexpr int NN = 20, CNT = (NN + 1) * NX, IT = (CNT + WAVE - 1) / WAVE;
    double xv[IT], yv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        int i = lane + WAVE * it; i = i < CNT ? i : CNT - 1;
        const int k = div7(i), c = i - 7 * k;
        xv[it] = LDG(xb + i);
        yv[it] = k < NN ? LDG(yr + k * 9 + c) : LDG(yre + c);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) { int i = lane + WAVE * it; i = i < CNT ? i : CNT - 1; dq[i] = xv[it] - yv[it]; }
}

// LDS map (doubles).  Per wave: what the interior point of its segment needs -- H (packed lower rows of Huu) / L (unit factor) in one
// buffer, Hb (the constant border rows: Hzu, Bbar), Lb (border rows of the factor), parked per-lane constants, exchange buffers (cb, invd, sb, sb2 in
// the relative layout dense40.h's col_head assumes), wv (border unknowns of the back substitution).  Phases A, C, E alias it as in
// admpc_fused20.hip.  Per segment an interface block (IF_*) that the other waves read; one word block per workgroup.
template <int S>
struct SegLds {
    static constexpr int N = 20, NTRI = 820;
    static constexpr int NB = S == 2 ? 7 : 14;                         // border rows per segment
    static constexpr int NR = 40 + NB;                                 // lanes that carry rows of the bordered matrix
    // H (needed from the top of an iteration to the row build of its factorisation) and L (from the factorisation to the last back
    // substitution) are never live together: ONE buffer; H waits in the workgroup's slot of global memory (L2) and is fetched at the top
    // of every iteration -- 6.6 KB of LDS per wave, the difference between three and four workgroups per CU at S = 2
    // S > 2 (14 border rows per wave): the constant border rows Hb and the factor's border rows Lb share their buffer the same way, Hb behind H in
    // the slot -- 4.5 KB per wave, the difference between one and two workgroups per CU at S = 4
    static constexpr bool share_b = S > 3;                                // (S = 3 has two workgroups per CU either way: measured slower with the extra fetch)
    static constexpr bool bl_if = S > 2;                                  // bl of the linearisation in the interface block (below)
    static constexpr int BS = 42;                                         // row stride of Hb / Lb (seg_cut.h)
    // LDS banks: a ds_read_b64 of a packed row start tri(i) = i (i + 1) / 2 is conflict-free over lanes 0..31 (triangular numbers are distinct mod 32) and
    // over rows 32..39 (bank pairs 16 17 19 22 26 31 5 12); the border rows in lanes 40.. share that half of the wave: with stride 42 (10 mod 32) and a base
    // that is 0 or 10 mod 32 doubles away from the packed rows, seven border rows sit on 0 10 20 30 8 18 28 (+ 10) -- clear of the rows above
    static constexpr int oHbRel = 832;                                    // Hb behind H (820), 0 mod 32
    static constexpr int NSLOT = share_b ? oHbRel + NB * BS : NTRI;       // doubles of a wave's slot (S = 4: H, the gap, Hb -- the LDS image)
    static constexpr int oH = 0, oL = oH, oHb = oH + oHbRel;
    static constexpr int oLb = share_b ? oHb : oHb + NB * BS + (((10 - (oHbRel + NB * BS)) % 32 + 32) % 32);
    static constexpr int oPark = oLb + NB * BS, oCb = oPark + 4 * 64;
    static constexpr int oWv = oCb + 4 * 64;
    static constexpr int JTS = 24, JTK = 4 * JTS + 2;
    static constexpr int oJT = 0, oBlA = N * JTK, oGTC = 0, oDqC = 860, oGam = oDqC + 148;
    // what the interior point needs / what phase A's tables need (S > 2: the residuals bl of the linearisation live in the interface block,
    // over fields only the interior point uses)
    static constexpr int seg_ipm = oWv + 16, seg_lin = oBlA + (bl_if ? 0 : N * NX);
    static constexpr int seg = seg_ipm > seg_lin ? seg_ipm : seg_lin;
    static_assert(seg % 2 == 0 && oGTC + N * GTS <= oDqC && oGam + (NX + 1) * 64 <= seg && oGam + (NX + 1) * 64 <= oCb, "LDS aliases");
    // interface block of a segment (the fields of the recursion over several cuts only exist for S > 2)
    static constexpr int SCS = NB;                                     // row stride of Sc
    static constexpr int IF_SC = 0;                                    // [NB][NB] C M^-1 C' of this iteration
    static constexpr int IF_HZZ = IF_SC + ((NB * NB + 1) & ~1);        // [7][8]   Hzz (cost part)
    static constexpr int IF_C = IF_HZZ + 56;                           // [8]      c_s = free response at the segment's end
    static constexpr int IF_ZB = IF_C + 8;                             // [16]     reduced right-hand side of the border rows (forward substitution)
    static constexpr int IF_Z = IF_ZB + 16;                            // [8]      z_s of the interior point's iterate
    static constexpr int IF_ZC = IF_Z + 8;                             // [8]      z_s of the cold start (the free chain)
    static constexpr int IF_DZ = IF_ZC + 8;                            // [8]      Newton step of z_s
    static constexpr int IF_RED = IF_DZ + 8;                           // [4][8]   partial reductions, one slot per barrier phase
    static constexpr int IF_BU = IF_RED + 32;                          // [8]      Bbar_s U_s + c_s of the returned inputs
    static constexpr int IF_G56 = IF_BU + 8;                           // [2]      sum of the delta-box barrier ratios of the segment
    static constexpr int IF_ABAR = IF_G56 + 2;                         // [7][8]   Abar_s = Phi at the segment's end            (S > 2)
    static constexpr int IF_PI = IF_ABAR + (S > 2 ? 56 : 0);           // [7][8]   Pi_s (backward recursion over the cuts)
    static constexpr int IF_X = IF_PI + 56;                            // [7][8]   X_s = Lambda_s^-1 Ahat_s                     (S > 2)
    static constexpr int IF_WI = IF_X + (S > 2 ? 56 : 0);              // [7][8]   Wi_s = Lambda_s^-1                           (S > 2)
    static constexpr int IFS = IF_WI + (S > 2 ? 56 : 0);
    static constexpr int IF_BL = IF_PI;                                // S > 2: bl [20][7] of phases A, C, E over Pi, X, Wi
    static_assert(!bl_if || IFS - IF_BL >= N * NX, "bl over the interior point's interface fields");
    static constexpr int oIF = S * seg, oWG = oIF + S * IFS;
    static constexpr int oYM = oWG + 8;                                // [4][7][8] solution operators of the single cut (S = 2)
    static constexpr int total = oYM + (S == 2 ? 4 * 56 : 0);
};

// ---- 7 x 7 blocks of the interface recursion (wave 0): seg_cut.h's column type, product and elimination with D = 7.  Matrices live in LDS,
// row-major; "lane = column": lane c holds column c of a matrix (or a right-hand side) in seven registers.
typedef ColD<7> Col7;
__device__ __forceinline__ Col7 mat_col(const double* A, const int sa, const int /*ka = 1*/, const Col7& x) { return cut_mat_col<7>(A, sa, x); }
__device__ __forceinline__ void ge7_solve(Col7& c) { cut_ge_solve<7>(c); }

// The coupling of the S segments through their S - 1 cuts.  Per segment s, from its interface block:
//   Pzz_s = Hzz_s + e6 e6' sum(G56) - Sc_zz,  Pzb_s = Sc_zb,  Pbb_s = Sc_bb,  Ahat_s = Abar_s - Pzb_s',
//   yhat_s = zb[z rows] (reduced stationarity of z_s),  dhat_s = -zb[Bbar rows]
// Backward over the cuts:  nu_s = eta_s - Pi_s dz_s  with  Pi_{S-1} = Pzz, eta_{S-1} = yhat;  Lambda_s = I + Pbb_s Pi_{s+1},
//   Pi_s = Pzz_s + Ahat_s' Pi_{s+1} Lambda_s^-1 Ahat_s,   eta_s = yhat_s + Ahat_s' (eta_{s+1} - Pi_{s+1} Lambda_s^-1 (dhat_s + Pbb_s eta_{s+1}))
// Forward:  dz_{s+1} = Lambda_s^-1 (Ahat_s dz_s + dhat_s + Pbb_s eta_{s+1}),  nu_{s+1} = eta_{s+1} - Pi_{s+1} dz_{s+1}.
// (tests/seg_spec.py: newton()).  What depends on the factorisation only -- Pi_s, X_s = Lambda_s^-1 Ahat_s, Wi_s = Lambda_s^-1 -- is built ONCE
// per interior-point iteration by wave 0 (interface_factor, while the other waves run their forward substitutions); every right-hand side then
// needs matrix-vector products only, which every wave runs for itself in registers (interface_apply: no barrier, the same bits in every wave).
template <int S>
__device__ __forceinline__ void interface_factor(double* __restrict__ ifb, const int lane) {
    using LD = SegLds<S>;
    const int l49 = lane < 49 ? lane : 48;
    const int er = div7(l49), ec = l49 - 7 * er;                    // element (er, ec) for the element-wise step
    // groups of eight lanes: columns of Lambda | of Ahat | of Lambda again (the DPP elimination wants the matrix in every 16-lane row) | of I
    const int g8 = lane >> 3, cl = lane & 7, c7 = cl < 7 ? cl : 0;
    const int grp = (g8 & 3) == 1 ? 1 : ((g8 & 3) == 3 ? 2 : 0);     // role: 0 = Lambda, 1 = Ahat, 2 = I
    const bool owner = g8 == 1 || g8 == 3;
    const bool keep = cl < 7 && owner;
    {                                                               // the last segment: Pi = Pzz
        double* F = ifb + (S - 1) * LD::IFS;
        const double v = F[LD::IF_HZZ + er * 8 + ec] + ((er == 6 && ec == 6) ? F[LD::IF_G56] : 0.0) - F[LD::IF_SC + er * LD::SCS + ec];
        if (lane < 49) F[LD::IF_PI + er * 8 + ec] = v;
    }
    WSYNC();
#pragma unroll 1
    for (int s = S - 2; s >= 0; --s) {
        double* F = ifb + s * LD::IFS;
        const double* Fn = ifb + (s + 1) * LD::IFS;
        const bool mid = s > 0;                                    // segment 0 has no z rows: its border rows 0..6 are Bbar, and nothing in front of it
        const double* Pbb = F + LD::IF_SC + (mid ? 7 * LD::SCS + 7 : 0);
        Col7 col;
#pragma unroll
        for (int r = 0; r < 7; ++r) col.v[r] = Fn[LD::IF_PI + r * 8 + c7];
        Col7 T = mat_col(Pbb, LD::SCS, 1, col);
        Col7 ah;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            ah.v[r] = mid ? F[LD::IF_ABAR + r * 8 + c7] - F[LD::IF_SC + c7 * LD::SCS + 7 + r] : 0.0;      // Ahat[r][c] = Abar[r][c] - Pzb[c][r]
            const double id = r == c7 ? 1.0 : 0.0;
            T.v[r] = grp == 0 ? T.v[r] + id : (grp == 1 ? ah.v[r] : id);
        }
        cut_ge_solve_dpp<7>(T);                                    // role 1: columns of X = Lambda^-1 Ahat, role 2: of Wi = Lambda^-1
        if (grp == 2 && keep) {
#pragma unroll
            for (int r = 0; r < 7; ++r) F[LD::IF_WI + r * 8 + cl] = T.v[r];
        }
        if (!mid) break;
        if (grp == 1 && keep) {
#pragma unroll
            for (int r = 0; r < 7; ++r) F[LD::IF_X + r * 8 + cl] = T.v[r];
        }
        Col7 Y = mat_col(Fn + LD::IF_PI, 8, 1, T);                 // group 1: Pi_{s+1} X
        if (grp == 1 && keep) {
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                double a = F[LD::IF_HZZ + r * 8 + cl] + ((r == 6 && cl == 6) ? F[LD::IF_G56] : 0.0) - F[LD::IF_SC + r * LD::SCS + cl];
#pragma unroll
                for (int k = 0; k < 7; ++k) a = fma(F[LD::IF_ABAR + k * 8 + r] - F[LD::IF_SC + r * LD::SCS + 7 + k], Y.v[k], a);      // Ahat[k][r]
                F[LD::IF_PI + r * 8 + cl] = a;
            }
        }
        WSYNC();
    }
}

// One right-hand side through the cuts (every wave for itself; lane r < 7 holds component r of every vector, products by DPP row broadcast):
// returns dz_w (0 for the first segment) and nu_{w+1} (0 for the last) of the wave's segment w.
template <int S>
__device__ __forceinline__ void interface_apply(const double* __restrict__ ifb, const int w, const int lane, double& dz_own, double& nu_next) {
    using LD = SegLds<S>;
    const int r = lane < 7 ? lane : 0;
    // y_r = sum_k M[r][k] x_k: the vector's component k comes from lane k of the 16-lane row inside the multiply-add (v_fmac_f64_dpp row_newbcast:k;
    // the first one carries the two wait states a DPP read needs behind the VALU write of x)
    auto mv = [&](const double* M, const int sr, const int sc, const double x) __attribute__((always_inline)) -> double {
        double m[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) m[k] = M[r * sr + k * sc];
        double a = 0.0;
        fmac_rowbc<0>(a, x, m[0]);
        static_for<1, 7>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; fmac_rowbc_ld<k>(a, x, m[k]); });
        return a;
    };
    double eta[S], xr[S], dz[S], nu[S];
    eta[S - 1] = ifb[(S - 1) * LD::IFS + LD::IF_ZB + r];
    static_for<1, S - 1>([&](auto ic) __attribute__((always_inline)) {
        constexpr int s = S - 1 - decltype(ic)::value;              // S-2 .. 1
        const double* F = ifb + s * LD::IFS;
        const double* Fn = ifb + (s + 1) * LD::IFS;
        const double v = mv(F + LD::IF_SC + 7 * LD::SCS + 7, LD::SCS, 1, eta[s + 1]) - F[LD::IF_ZB + 7 + r];
        xr[s] = mv(F + LD::IF_WI, 8, 1, v);
        const double wv = eta[s + 1] - mv(Fn + LD::IF_PI, 8, 1, xr[s]);
        double e = F[LD::IF_ZB + r];
        {
            double m[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) m[k] = F[LD::IF_ABAR + k * 8 + r] - F[LD::IF_SC + r * LD::SCS + 7 + k];      // Ahat[k][r]: Ahat' wv
            fmac_rowbc<0>(e, wv, m[0]);
            static_for<1, 7>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; fmac_rowbc_ld<k>(e, wv, m[k]); });
        }
        eta[s] = e;
    });
    {
        const double v = mv(ifb + LD::IF_SC, LD::SCS, 1, eta[1]) - ifb[LD::IF_ZB + r];
        dz[1] = mv(ifb + LD::IF_WI, 8, 1, v);
    }
    dz[0] = 0.0; nu[0] = 0.0;
    static_for<1, S>([&](auto sc_) __attribute__((always_inline)) {
        constexpr int s = decltype(sc_)::value;
        const double* F = ifb + s * LD::IFS;
        if constexpr (s > 1) dz[s] = mv(ifb + (s - 1) * LD::IFS + LD::IF_X, 8, 1, dz[s - 1]) + xr[s - 1];
        nu[s] = eta[s] - mv(F + LD::IF_PI, 8, 1, dz[s]);
    });
    dz_own = 0.0; nu_next = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) { if (w == s) { dz_own = dz[s]; if (s + 1 < S) nu_next = nu[s + 1]; } }
}

// Two segments, one cut: the coupling has a closed form in four 7 x 7 operators that depend on the factorisation only (seg_cut.h:
// cut_operators2 / cut_apply2), built ONCE per interior-point iteration by wave 0 while wave 1 already runs its forward substitution.
__device__ __forceinline__ void interface_factor2(double* __restrict__ ifb, double* __restrict__ YM, const int lane) {
    using LD = SegLds<2>;
    double* F1 = ifb + LD::IFS;
    cut_operators2<7, 8, true>(ifb + LD::IF_SC, LD::SCS, F1 + LD::IF_HZZ, 8, F1[LD::IF_G56], 6, F1 + LD::IF_SC, LD::SCS, F1 + LD::IF_PI, YM, lane);
}

template <int S, int QMASK>
__global__ __launch_bounds__(WAVE * S, 2) void admpc_seg_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                                const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                                const double* __restrict__ yrefeg, const double* __restrict__ pg,
                                                                double* __restrict__ xbarg, double* __restrict__ ubarg,
                                                                double* __restrict__ costg, int32_t* __restrict__ statusg,
                                                                int32_t* __restrict__ itersg, int first_pass, int* __restrict__ sched, int cap,
                                                                double* __restrict__ hslot)
{
    using LD = SegLds<S>;
    constexpr int N = 20, n = 40, NT = N * S, NB = LD::NB, NR = LD::NR, BS = LD::BS;
    extern __shared__ double lds_raw[];
    const int wv_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // segment of this wave (uniform)
    const bool first = wv_ == 0, last = wv_ == S - 1;
    const int k0 = wv_ * N;                                      // first stage of the segment
    const int bslot = first ? 0 : 7;                             // border rows [bslot, bslot + 7) hold Bbar (the first segment has no z rows)
    double* const lds_seg = lds_raw + wv_ * LD::seg;
    double* const ifb = lds_raw + LD::oIF;                       // interface blocks of all segments
    double* const IFm = ifb + wv_ * LD::IFS;                     // this segment's
    int* const wgw = reinterpret_cast<int*>(lds_raw + LD::oWG);  // workgroup words: [0] instance of this round
    double* const hsl = hslot + ((size_t)blockIdx.x * S + wv_) * LD::NSLOT;      // this wave's H in global memory (L2), rewritten per instance
    double* const YM = lds_raw + LD::oYM;                        // S = 2: the cut's solution operators (interface_factor2)
    double* const Hp = lds_seg + LD::oH;
    double* const Hb = lds_seg + LD::oHb;
    double* const Lp = lds_seg + LD::oL;
    double* const Lb = lds_seg + LD::oLb;
    double* const park = lds_seg + LD::oPark;
    double* const cb = lds_seg + LD::oCb;
    double* const invd = cb + 64;
    double* const sb = invd + 64;
    double* const sb2 = sb + 64;
    double* const wvec = lds_seg + LD::oWv;
    double* const JT = lds_seg + LD::oJT;
    double* const GT = lds_seg + LD::oGTC;
    double* const bl = LD::bl_if ? IFm + LD::IF_BL : lds_seg + LD::oBlA;
    double* const dqC = lds_seg + LD::oDqC;
    double* const gam = lds_seg + LD::oGam;
    double* const dqE = lds_seg + LD::oDqC;
    double* const dus = lds_seg + LD::oGam;
#define PK_DL   park[0 * 64 + lane]
#define PK_DUU  park[1 * 64 + lane]
#define PK_G0   park[2 * 64 + lane]
#define PK_DDL  park[3 * 64 + (lane & 31)]           // the steering box: stages = lanes < 20 (the two bounds share one row)
#define PK_DDU  park[3 * 64 + 32 + (lane & 31)]
#define LAUNDER_LANE(v) int v = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(v))
#define LAUNDER_CFG(c) int c##_z = 0; asm volatile("" : "+s"(c##_z)); const AdmpcConfig* __restrict__ c = cfg + c##_z
    // partial reductions of the S waves: slot `ph` of every segment, combined in segment order by every wave (same bits everywhere)
#define RED(s_, ph_, i_) ifb[(s_) * LD::IFS + LD::IF_RED + (ph_) * 8 + (i_)]

    SEG_DECL();
    // LDS byte address of the row this lane holds: packed row of H / L (lanes < 40), border row (lanes 40 .. NR-1), else row 0
    auto row_addr = [&](double* tri, double* brd, const int lz_) __attribute__((always_inline)) -> unsigned {
        const int lb = lz_ - n;
        return lds_byte_addr(lz_ < n ? tri + lz_ * (lz_ + 1) / 2 : (lz_ < NR ? brd + lb * BS : tri));
    };

    // Factorisation of the bordered Newton matrix: M = H + diag(dbar) + (s_odd on the odd columns of the u1 rows) = L D L' with the
    // border rows (lanes 40 .. NR-1: Hb, plus the delta-box barrier on the z6 row when zbar) riding along -> Lb = C L^-T D^-1.
    const Dense40bLds W{Hp, Hb, Lp, Lb, cb, invd};
    auto factorise = [&](const double dbar_, const double sodd_, const int lz_, const bool zbar, const double h_) __attribute__((always_inline)) {
        dense40b_factorise<NR, BS>(W, dbar_, lz_, sodd_, [&](double (&a)[n]) __attribute__((always_inline)) {
            // H's rows are in registers: its buffer becomes the factor's.  Diagonal slots of the packed factor: 0.0 (the factorisation stores the
            // strictly-lower part only; the substitution assembly lets the source lane of a step take part with this multiplier)
            if (lz_ < n) Lp[lz_ * (lz_ + 1) / 2 + lz_] = 0.0;
            if (zbar) {
                // row z6 (lane 46) of Qzu: + h * (sum of the barrier ratios of the stages behind the column's stage), odd columns; sb2[k] holds that sum
                const double hz = lz_ == 46 ? h_ : 0.0;
                static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    a[2 * k + 1] = fma(hz, sb2[k], a[2 * k + 1]);
                });
            }
        });
    };
    // Schur blocks of the border: Sc = Lb D Lb' (NB x NB) in ten v_mfma_f64_16x16x4_f64 (seg_cut.h)
    auto schur = [&](const int lane) __attribute__((always_inline)) { dense40b_schur<NB, BS>(W, IFm + LD::IF_SC, LD::SCS, lane); };
    // The coupled Newton solve for one right-hand side: y on the input lanes, the reduced stationarity of z_s on the z lanes (0 on
    // the Bbar lanes).  Returns the step of this lane's input; dz_s / nu_{s+1} are in the interface blocks afterwards.
    auto coupled_solve = [&](double y, const int lz_) __attribute__((always_inline)) -> double {
        const bool uz_ = lz_ < n;
        const unsigned pub = lds_byte_addr(cb + (lz_ & 15));
        fwd_subst_40_b<NR>(y, row_addr(Lp, Lb, lz_), pub);
        if (lz_ >= n && lz_ < NR) IFm[LD::IF_ZB + lz_ - n] = y;
        SEG_STAMP(6);
        XSYNC();
        if constexpr (S == 2) {
            // one cut: both waves apply the operators of interface_factor2 to (dhat, eta) themselves -- wave 0 needs nu_1 (its border rows
            // are Bbar: unknown -nu_1), wave 1 needs dz_1 (its border rows are the z rows)
            const double a = cut_apply2<7, 8>(YM, first, ifb + LD::IF_ZB, ifb + LD::IFS + LD::IF_ZB, lz_);
            if (lz_ < 7) { wvec[lz_] = first ? -a : a; if (!first) IFm[LD::IF_DZ + lz_] = a; }
            SEG_STAMP(7);
        } else {
            double dzo, nun;
            interface_apply<S>(ifb, wv_, lz_, dzo, nun);
            // unknowns of the border rows: dz_s on the z rows, -nu_{s+1} on the Bbar rows
            if (lz_ < 7) {
                wvec[lz_] = first ? -nun : dzo;
                wvec[7 + lz_] = (first || last) ? 0.0 : -nun;
                if (!first) IFm[LD::IF_DZ + lz_] = dzo;
            }
            SEG_STAMP(7);
        }
        WSYNC();
        double x = y * invd[uz_ ? lz_ : 0];
        {
            const int li = uz_ ? lz_ : 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) x = fma(-Lb[b * BS + li], wvec[b], x);
        }
        bwd_subst_40(x, lds_byte_addr(Lp + (uz_ ? lz_ : 0)), pub);
        SEG_STAMP(8);
        return x;
    };

    bool first_ticket = true;
    for (;;) {
        {
            LAUNDER_LANE(lane0);
            if (wv_ == 0) {
                const int t_ = f20_next(sched, cap, first_ticket, lane0);
                if (lane0 == 0) wgw[0] = t_;
            }
            first_ticket = false;
            XSYNC();
        }
        const int inst = __builtin_amdgcn_readfirstlane(wgw[0]);
        if (inst < 0) break;
        if (!first_pass && statusg[inst] != 0) { XSYNC(); continue; }      // failed / converged in an earlier SQP iteration of this call
        double* const xbg = xbarg + (size_t)inst * (NT + 1) * NX;
        double* const ubg = ubarg + (size_t)inst * NT * NU;
        const double* yrg = yrefg + (size_t)inst * NT * NY;
        double* const xbs = xbg + (size_t)k0 * NX;                  // the segment's rows
        double* const ubs = ubg + (size_t)k0 * NU;
        const double* yrs = yrg + (size_t)k0 * NY;
        const double* yre = last ? yrefeg + (size_t)inst * NX : yrs + N * NY;      // row 20 of the segment's references (unused unless last)

        SEG_STAMP(0);
#ifdef ADMPC_PHASE_TIMERS
        const unsigned long long tr_t0 = seg_real(); unsigned long long tr_t1 = 0;
#endif
        double du = 0.0;
        bool failed = false;
        int it = 0;
        int npass = 2; asm volatile("" : "+s"(npass));
#pragma unroll 1
        for (int pass = 0; pass < npass; ++pass) {
        // =================================================================================================================
        // phase A (H0/H1): ERK4 + forward sensitivities of the segment's 20 stages (text of admpc_fused20.hip, stage offset k0)
        // =================================================================================================================
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const double h = cf->Ts;
            const int tsk = lane < 3 * N ? lane : 3 * N - 1;
            const int k = (int)(((unsigned)tsk * 21846u) >> 16), g = tsk - 3 * k;
            const bool live = lane < 3 * N;
            {
                const double pin = pg[inst];
                double x[NX], u[NU], xn1[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) { x[i] = LDG(xbs + k * NX + i); xn1[i] = LDG(xbs + (k + 1) * NX + i); }
                u[0] = LDG(ubs + k * NU); u[1] = LDG(ubs + k * NU + 1);
                double kx[NX], accx[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) { kx[i] = 0.0; accx[i] = 0.0; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double cs = (s == 0) ? 0.0 : (s == 3 ? 1.0 : 0.5);
                    const double ws = (s == 0 || s == 3) ? (1.0 / 6.0) : (2.0 / 6.0);
                    double X[NX];
#pragma unroll
                    for (int i = 0; i < NX; ++i) X[i] = x[i] + cs * h * kx[i];
                    ModelEvalT<double> e;
                    model_eval<double>(cf, X, u, pin, e);
#pragma unroll
                    for (int i = 0; i < NX; ++i) { kx[i] = e.f[i]; accx[i] += ws * e.f[i]; }
                    if (live && g == 0) {
                        double2* jt = reinterpret_cast<double2*>(JT + k * LD::JTK + s * LD::JTS);
                        jt[0] = make_double2(e.j0[0], e.j0[1]); jt[1] = make_double2(e.j0[2], e.j1[0]); jt[2] = make_double2(e.j1[1], e.j1[2]);
#pragma unroll
                        for (int r = 0; r < 3; ++r) { jt[3 + 2 * r] = make_double2(e.a[r][0], e.a[r][1]); jt[4 + 2 * r] = make_double2(e.a[r][2], e.a[r][3]); }
#pragma unroll
                        for (int r = 0; r < 3; ++r) jt[9 + r] = make_double2(e.bu[r][0], e.bu[r][1]);
                    }
                }
                if (live && g == 0) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) bl[k * NX + i] = (x[i] + h * accx[i]) - xn1[i];
                }
            }
            WSYNC();
            {
                const int xcol0 = g == 0 ? 2 : 5;
                double kS[3][NX], accS[3][NX];
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                    for (int i = 0; i < NX; ++i) { kS[cc][i] = 0.0; accS[cc][i] = 0.0; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double cs = (s == 0) ? 0.0 : (s == 3 ? 1.0 : 0.5);
                    const double ws = (s == 0 || s == 3) ? (1.0 / 6.0) : (2.0 / 6.0);
                    double Sx[3][NX];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                        for (int i = 0; i < NX; ++i) {
                            const double id = (g < 2 && i == xcol0 + cc) ? 1.0 : 0.0;
                            Sx[cc][i] = id + cs * h * kS[cc][i];
                        }
                    ModelEvalT<double> e;
                    {
                        const double2* jt = reinterpret_cast<const double2*>(JT + k * LD::JTK + s * LD::JTS);
                        double2 q;
                        q = jt[0]; e.j0[0] = q.x; e.j0[1] = q.y; q = jt[1]; e.j0[2] = q.x; e.j1[0] = q.y; q = jt[2]; e.j1[1] = q.x; e.j1[2] = q.y;
#pragma unroll
                        for (int r = 0; r < 3; ++r) { q = jt[3 + 2 * r]; e.a[r][0] = q.x; e.a[r][1] = q.y; q = jt[4 + 2 * r]; e.a[r][2] = q.x; e.a[r][3] = q.y; }
#pragma unroll
                        for (int r = 0; r < 3; ++r) { q = jt[9 + r]; e.bu[r][0] = q.x; e.bu[r][1] = q.y; }
                    }
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        sens_rhs<double>(e, Sx[cc], g == 2 ? cc : -1, kS[cc]);
#pragma unroll
                        for (int i = 0; i < NX; ++i) accS[cc][i] += ws * kS[cc][i];
                    }
                }
                const int c0 = g == 0 ? 0 : (g == 1 ? 3 : 5);
                const int nc = g == 0 ? 3 : 2;
                WSYNC();                                   // GT overlays the Jacobian tables: every lane has read its tables before the first store
                double* Gk = GT + k * GTS;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    if (live && cc < nc)
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            const double id = (g < 2 && i == xcol0 + cc) ? 1.0 : 0.0;
                            Gk[(c0 + cc) * 6 + i] = id + h * accS[cc][i];
                        }
            }
            WSYNC();
        }
        if (pass != 0) { SEG_STAMP(11); break; }
        SEG_STAMP(1);

        // =================================================================================================================
        // phase C (H2-H4): condensing of the segment.  Lane i < 40 <-> input i = 2k + j carries column i of Gamma_k = d x_k / d U_s,
        // lanes 40..46 (segments behind the first) the columns of Phi_k = d x_k / d z_s; all lanes the free response xhat_k
        // (from z_s = 0; the first segment: from dx_0).  H = [Huu Huz; Hzu Hzz] on six v_mfma_f64_16x16x4_f64 tiles.
        // =================================================================================================================
        double g0, xh6_own = 0.0, rsx = 0.0;
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const int ki = lane >> 1, ji = lane & 1;
            const bool uact = lane < n;
            const int sc = uact ? lane : 0;
            const double Ts = cf->Ts, h = cf->Ts;
            const double Rj = Ts * cf->W[NX + ji];
            stage_dq_seg(dqC, xbs, yrs, yre, lane);
            const double ubar_i = ubs[sc];
            const double r_i = uact ? Rj * (ubar_i - yrs[(sc >> 1) * 9 + 7 + (sc & 1)]) : 0.0;
            double Qd[NX], Qe[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { Qd[i] = Ts * cf->W[i]; Qe[i] = cf->We[i]; }
            // the carried columns in two sets used alternately (stage k reads set k & 1 and writes the other one, see admpc_fused20.hip)
            double xh2[2][NX], g2[2][NX];
#pragma unroll
            for (int c = 0; c < NX; ++c) { xh2[0][c] = first ? x0g[(size_t)inst * NX + c] - xbg[c] : 0.0; xh2[1][c] = 0.0; }      // uniform
            WSYNC();
#pragma unroll
            for (int c = 0; c < NX; ++c) { g2[0][c] = (!first && lane == n + c) ? 1.0 : 0.0; g2[1][c] = 0.0; }
            constexpr int NCOMP = ((QMASK >> 0) & 1) + ((QMASK >> 1) & 1) + ((QMASK >> 2) & 1) + ((QMASK >> 3) & 1) + ((QMASK >> 4) & 1) + ((QMASK >> 5) & 1) + ((QMASK >> 6) & 1);
            constexpr int NSTEP = (NCOMP + 3) / 4;
            const int r16 = lane & 15, kq = lane >> 4;
            int crow[NSTEP]; double wq_l[NSTEP], we_l[NSTEP];
            {
                int seen = 0;
#pragma unroll
                for (int st_ = 0; st_ < NSTEP; ++st_) { crow[st_] = 7; wq_l[st_] = 0.0; we_l[st_] = 0.0; }
                static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr ((QMASK >> c) & 1) {
                        const int st_ = seen >> 2, kk = seen & 3;
#pragma unroll
                        for (int q_ = 0; q_ < NSTEP; ++q_) if (q_ == st_ && kq == kk) { crow[q_] = c; wq_l[q_] = Qd[c]; we_l[q_] = Qe[c]; }
                        ++seen;
                    }
                });
            }
            gam[7 * 64 + lane] = 0.0;
            d4 acc[3][3];
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J < 3; ++J) acc[I][J] = d4{0.0, 0.0, 0.0, 0.0};
            g0 = r_i;
            static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int lim = 2 * k < n ? 2 * k : n;
                constexpr int nblk = (lim + 15) / 16;             // 16-lane blocks of Gamma that are non-zero at this stage (first segment)
                double (&g)[NX] = g2[k & 1]; double (&xh)[NX] = xh2[k & 1];
                double (&gn)[NX] = g2[(k + 1) & 1]; double (&xn)[NX] = xh2[(k + 1) & 1];
                int tok = B; asm volatile("" : "+s"(tok));         // one stage = one basic block (see admpc_fused20.hip)
                if (tok > 0) {
                // the tracking cost of stage k0 + k belongs to this segment unless it is the fixed first stage of the horizon or the
                // first stage of the next segment; stage NT carries the terminal weights
                const bool cost_k = k == 0 ? !first : (k == N ? last : true);
                double wg[NX];
                double blk[NSTEP][3];
#if SEG_CPREF
                // every LDS read of the stage's record is issued before anything else of the stage (see admpc_fused20.hip)
                double2 Av[15], Bv[3]; double blv[NX];
                if constexpr (k < N) {
                    const double* Gk = GT + k * GTS;
                    const double* const bsrc = ki == k ? Gk + 5 * 6 + 6 * ji : gam + 7 * 64;
#pragma unroll
                    for (int q_ = 0; q_ < 15; ++q_) Av[q_] = *reinterpret_cast<const double2*>(Gk + 2 * q_);
#pragma unroll
                    for (int q_ = 0; q_ < 3; ++q_) Bv[q_] = *reinterpret_cast<const double2*>(bsrc + 2 * q_);
#pragma unroll
                    for (int r = 0; r < NX; ++r) blv[r] = bl[k * 7 + r];
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
                if constexpr (k < N) { if (lane == k) xh6_own = xh[6]; }
                if (cost_k) {
                    static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr ((QMASK >> c) & 1) {
                            const double w = k < N ? Qd[c] : Qe[c];
                            wg[c] = w * g[c];
                            g0 += wg[c] * (xh[c] + dqC[k * 7 + c]);
                            gam[c * 64 + lane] = g[c];
                        }
                    });
#pragma unroll
                    for (int st_ = 0; st_ < NSTEP; ++st_)
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            if (m < nblk || !first) blk[st_][m] = gam[crow[st_] * 64 + 16 * m + r16];
                            else blk[st_][m] = 0.0;
                        }
                }
                if constexpr (k < N) {
                    const double* Gk = GT + k * GTS;
                    // The inputs of stage k enter with B_k: lanes 2k, 2k + 1 (whose column of Gamma is still zero) start the product from
                    // their column of B_k, every other lane from the zero row of gam -- one per-lane LDS address instead of 28 selects
                    // per stage (560 vector instructions per instance); 0 + A_k 0 = 0 exactly: the same bits as the selects gave.
                    const bool mine = ki == k;
#if SEG_CPREF
#pragma unroll
                    for (int r = 0; r < 6; r += 2) { gn[r] = Bv[r / 2].x; gn[r + 1] = Bv[r / 2].y; }
#else
                    const double* const bsrc = mine ? Gk + 5 * 6 + 6 * ji : gam + 7 * 64;
#pragma unroll
                    for (int r = 0; r < 6; r += 2) {
                        const double2 v = *reinterpret_cast<const double2*>(bsrc + r);
                        gn[r] = v.x; gn[r + 1] = v.y;
                    }
                    double blv[NX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) blv[r] = bl[k * 7 + r];
#endif
                    gn[0] += g[0]; gn[1] += g[1];
                    gn[6] = mine ? (ji ? h : 0.0) : g[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) xn[r] = r < 2 ? blv[r] + xh[r] : blv[r];
                    xn[6] = blv[6] + xh[6];
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
#pragma unroll
                        for (int r = 0; r < 6; r += 2) {
#if SEG_CPREF
                            const double2 a = Av[c * 3 + r / 2];
#else
                            const double2 a = *reinterpret_cast<const double2*>(Gk + c * 6 + r);
#endif
                            xn[r] += a.x * xh[c + 2]; xn[r + 1] += a.y * xh[c + 2];
                            gn[r] += a.x * g[c + 2];  gn[r + 1] += a.y * g[c + 2];
                        }
                    }
                }
                if (cost_k) {
#pragma unroll
                    for (int st_ = 0; st_ < NSTEP; ++st_) {
                        const double wl = k < N ? wq_l[st_] : we_l[st_];
                        static_for<0, 3>([&](auto Ic) __attribute__((always_inline)) {
                            constexpr int I = decltype(Ic)::value;
                            const double a = wl * blk[st_][I];
                            static_for<0, I + 1>([&](auto Jc) __attribute__((always_inline)) {
                                constexpr int J = decltype(Jc)::value;
                                if constexpr (I < nblk) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, blk[st_][J], acc[I][J], 0, 0, 0);
                                else { if (!first) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, blk[st_][J], acc[I][J], 0, 0, 0); }
                            });
                        });
                    }
                }
                }
            });
            double (&g)[NX] = g2[N & 1]; double (&xh)[NX] = xh2[N & 1];        // what the last stage left
            // ---- the cut behind the segment: Abar_s (columns in lanes 40..46), c_s; then the free chain z^cold of the cold start
            if (!last) {
                if (!first && lane >= n && lane < n + 7) {
#pragma unroll
                    for (int r = 0; r < 7; ++r) IFm[LD::IF_ABAR + r * 8 + lane - n] = g[r];
                }
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < 7; ++r) IFm[LD::IF_C + r] = xh[r];
                }
            }
            XSYNC();
            // z^cold_1 = c_0, z^cold_{t+1} = Abar_t z^cold_t + c_t: lanes 0..6 = components (every wave runs the chain up to its own cut)
            double zc = 0.0;
            if (!first) {
                const int r7 = lane < 7 ? lane : 0;
                zc = ifb[LD::IF_C + r7];
#pragma unroll 1
                for (int t = 1; t < wv_; ++t) {
                    cb[lane] = zc;
                    WSYNC();
                    const double* Ft = ifb + t * LD::IFS;
                    double a = Ft[LD::IF_C + r7];
#pragma unroll
                    for (int c = 0; c < 7; ++c) a = fma(Ft[LD::IF_ABAR + r7 * 8 + c], cb[c], a);
                    WSYNC();
                    zc = a;
                }
                if (lane < 7) { IFm[LD::IF_ZC + lane] = zc; IFm[LD::IF_Z + lane] = zc; }
            }
            // ---- stationarity rows of the states at the cold start, | w (dx_k + xbar_k - xref_k) | over the segment's cost stages
            // (component 6 of the stages that carry a delta box is evaluated with its multipliers in phase D): the start value of the
            // tracked residual of the stopping test (oracle ipm_solve: rstat; rowqp_core.h: pass_init).  Lane r < 7 <-> component r,
            // dx through the linearised dynamics with zero inputs (text of the expansion, phase E).
            {
                const int r6 = lane < 6 ? lane : 0, r7 = lane < NX ? lane : 0;
                const double wq = lane < NX ? Ts * cf->W[r7] : 0.0, wqe = lane < NX ? cf->We[r7] : 0.0;
                double dx = lane < NX ? (first ? x0g[(size_t)inst * NX + r7] - xbg[r7] : zc) : 0.0;
                double m = 0.0;
                static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    const bool cost_k = k == 0 ? !first : (k == N ? last : true);
                    const double e = dx + dqC[k * 7 + r7];
                    const bool row_here = lane < NX && cost_k && (k == N || lane != 6);
                    m = fmax(m, row_here ? fabs((k < N ? wq : wqe) * e) : 0.0);
                    if constexpr (k < N) {
                        const double* Gk = GT + k * GTS;
                        const double bk = bl[k * 7 + r7];
                        double a = bk + (lane < 2 ? dx : 0.0);
                        double gg[5];
#pragma unroll
                        for (int c = 0; c < 5; ++c) gg[c] = Gk[c * 6 + r6];
                        fmac_rowbc<2>(a, dx, gg[0]); fmac_rowbc<3>(a, dx, gg[1]); fmac_rowbc<4>(a, dx, gg[2]);
                        fmac_rowbc<5>(a, dx, gg[3]); fmac_rowbc<6>(a, dx, gg[4]);
                        dx = lane < 6 ? a : (lane == 6 ? bk + dx : 0.0);
                    }
                });
                rsx = wave_reduce<OpMaxNan>(m);
            }
            WSYNC();
            // ---- the tiles into the packed rows: rows < 40 -> H (lower triangle), rows 40..46 -> Hzu (border rows 0..6) and Hzz
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * I + kq + 4 * v, col = 16 * J + r16;
                        if (row < n && col <= row) Hp[row * (row + 1) / 2 + col] = acc[I][J][v];
                        if (I == 2 && !first && row >= n && row < n + 7) {
                            if (col < n) Hb[(row - n) * BS + col] = acc[I][J][v];
                            else if (col < n + 7) IFm[LD::IF_HZZ + (row - n) * 8 + col - n] = acc[I][J][v];
                        }
                    }
            // Bbar_s: column i of Gamma at the segment's end is in lane i -> border rows bslot .. bslot + 6; unused border rows: zero
            if (uact) {
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    if (!last) Hb[(bslot + r) * BS + lane] = g[r];
                    if (NB == 14 && (first || last)) Hb[(7 + r) * BS + lane] = 0.0;
                }
            }
            WSYNC();
            // H to the wave's slot: the factor of the trial takes its buffer.  The vector L1 does not follow the wave's own stores and may
            // still hold lines the PREVIOUS instance of this wave read here: drop them once the stores have retired.
            stage_in<LD::NTRI>(hsl, Hp, lane);
            if constexpr (LD::share_b) { stage_in<7 * BS>(hsl + LD::oHbRel, Hb, lane); stage_in<7 * BS>(hsl + LD::oHbRel + 7 * BS, Hb + 7 * BS, lane); }      // Hb behind it (Lb takes its buffer)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // invalidate only: a release would write the L2's dirty lines back to HBM
            WSYNC();
        }

        SEG_STAMP(2);
#ifdef SEG_DEBUG
        if (inst == g_seg_dbg_inst) {
            LAUNDER_LANE(ld_);
            double* D = g_seg_dbg + (size_t)(wv_ * 2 + 0) * SEG_DBG_W;
            seg_dbg_copy(D, lds_seg, LD::seg, ld_);
            seg_dbg_copy(D + 4000, IFm, LD::IFS, ld_);
            D[5000 + ld_] = g0; D[5064 + ld_] = xh6_own; D[5128 + ld_] = rsx;
        }
#endif
        // =================================================================================================================
        // phase D (H5): unconstrained trial + interior point, segments coupled through the cuts
        // =================================================================================================================
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const int ki = lane >> 1, ji = lane & 1;
            const bool uact = lane < n;
            const bool dact = lane < N && (lane >= 1 || !first);
            const bool zact = !first && lane >= n && lane < n + 7;
            const int sc = uact ? lane : 0;
            const double Ts = cf->Ts, h = cf->Ts;
            const double Rj = Ts * cf->W[NX + ji];
            const double rho_l = Ts * cf->zl, rho_u = Ts * cf->zu;
            const double thr = cf->ipm_thr0, mu0 = cf->ipm_mu0;
            const double tol_comp = cf->ipm_tol_comp, tol_res = cf->ipm_tol_res, tol_step = cf->ipm_tol_step;
            const int itmax = cf->ipm_iter_max;
            const bool try_unc = cf->ipm_try_unconstrained != 0.0;
            const double thw = cf->ipm_warm_thr, wrest = cf->ipm_warm_restart;
            const int fbit = (int)cf->ipm_fallback_iter;
            const double inv_nineq = 1.0 / (double)(8 * NT + 2 * (NT - 1));
            const double ubar_i = ubs[sc];
            const double dl_i = cf->lbu[ji] - ubar_i, duu_i = cf->ubu[ji] - ubar_i;
            const double z6c = first ? 0.0 : uni(IFm[LD::IF_ZC + 6]);
            double t[4], lam[4], sl = thr, su = thr;
            {
                const double r0[4] = { thr - dl_i, thr + duu_i, thr, thr };
#pragma unroll
                for (int i = 0; i < 4; ++i) { t[i] = r0[i] > thr ? r0[i] : thr; lam[i] = mu0 * rcp_nr(t[i]); }
            }
            double Dt[2] = {1.0, 1.0}, Dlam[2] = {0.0, 0.0}, Ddl = 0.0, Ddu = 0.0, dx6 = 0.0;
            if (dact) {
                const double x6 = xbs[lane * 7 + 6];
                Ddl = cf->lbx_delta - x6; Ddu = cf->ubx_delta - x6;
                dx6 = z6c + xh6_own;
                const double r0[2] = { dx6 - Ddl, Ddu - dx6 };
#pragma unroll
                for (int i = 0; i < 2; ++i) { Dt[i] = r0[i] > thr ? r0[i] : thr; Dlam[i] = mu0 * rcp_nr(Dt[i]); }
            }
            PK_DL = dl_i; PK_DUU = duu_i; PK_G0 = g0; if (lane < 32) { PK_DDL = Ddl; PK_DDU = Ddu; }
            WSYNC();

            // z part of the cost gradient at the current z_s (segments behind the first): the input lanes get Hzu' z, the z lanes Hzz z
            auto zgrad = [&](const int lz_) __attribute__((always_inline)) -> double {
                double a = 0.0;
                if (!first) {
                    const bool uz_ = lz_ < n;
                    const double* base = uz_ ? Hb + lz_ : IFm + LD::IF_HZZ + ((lz_ >= n && lz_ < n + 7) ? lz_ - n : 0) * 8;
                    const int stride = uz_ ? BS : 1;
#pragma unroll
                    for (int c = 0; c < 7; ++c) a = fma(base[c * stride], IFm[LD::IF_Z + c], a);
                }
                return a;
            };

            double rmax_prev = 0.0, step = 1e300, stp_local = 1e300, alpha_prev = 1.0, rstat = -1.0;
            bool solved = false, warmed = false, cons = false;
            bool hres = false;                                           // H (and Hb) resident or on its way from the slot
            if (try_unc) {
                int lt = lane; asm volatile("" : "+v"(lt));
                const double zg = zgrad(lt);                                 // reads Hb: in front of the factorisation (S > 2: Lb takes Hb's buffer)
                factorise(uact ? Rj : 1.0, 0.0, lt, false, h);
                schur(lt); if (lt == 0) IFm[LD::IF_G56] = 0.0;
                XSYNC(); if (wv_ == 0) { if constexpr (S == 2) interface_factor2(ifb, YM, lt); else interface_factor<S>(ifb, lt); }
                const double xt = coupled_solve(uact ? -(g0 + zg) : (zact ? -(g0 + zg) : 0.0), lt);
                slot_fetch<LD::NSLOT>(Hp, hsl, lt); hres = true;          // the trial's factor is dead
                const double duc = uact ? xt : 0.0;
#ifdef SEG_DEBUG
                if (inst == g_seg_dbg_inst) {
                    double* D = g_seg_dbg + (size_t)(wv_ * 2 + 1) * SEG_DBG_W;
                    seg_dbg_copy(D, lds_seg, LD::seg, lane);
                    seg_dbg_copy(D + 4000, IFm, LD::IFS, lane);
                    D[5000 + lane] = duc; D[5064 + lane] = zg; D[5128 + lane] = g0;
                }
#endif
                cb[lane] = duc;
                WSYNC();
                const double dz6 = first ? 0.0 : IFm[LD::IF_DZ + 6];
                const double du1_stage = lane < N ? cb[2 * lane + 1] : 0.0;
                const double pre = wave_scan_incl<OpSum>(du1_stage);
                const double dx6c = z6c + dz6 + xh6_own + h * (pre - du1_stage);
                const bool ok = (!uact || (duc >= dl_i && duc <= duu_i)) && (!dact || (dx6c >= Ddl && dx6c <= Ddu));
                WSYNC();
                const bool wok = __all(ok);
                if (lane == 0) RED(wv_, 1, 0) = wok ? 1.0 : 0.0;
                XSYNC();
                bool allok = true;
#pragma unroll
                for (int s = 0; s < S; ++s) allok = allok && RED(s, 1, 0) != 0.0;
                if (allok) { du = duc; solved = true; }
                else if (thw > 0.0) {
                    warmed = true;
                    du = duc;
                    if (!first && lane < 7) IFm[LD::IF_Z + lane] += IFm[LD::IF_DZ + lane];
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
                    WSYNC();
                }
            }
            SEG_STAMP(3);
#ifdef ADMPC_PHASE_TIMERS
            tr_t1 = seg_real();
#endif
            auto cold_start = [&]() __attribute__((always_inline)) {
                const double dlc = PK_DL, duc2 = PK_DUU;
                du = 0.0; sl = thr; su = thr;
                const double r0[4] = { thr - dlc, thr + duc2, thr, thr };
#pragma unroll
                for (int i = 0; i < 4; ++i) { t[i] = r0[i] > thr ? r0[i] : thr; lam[i] = mu0 * rcp_nr(t[i]); }
                if (!first && lane < 7) IFm[LD::IF_Z + lane] = IFm[LD::IF_ZC + lane];
                dx6 = dact ? z6c + xh6_own : 0.0;
                const double q0[2] = { dx6 - PK_DDL, PK_DDU - dx6 };
#pragma unroll
                for (int i = 0; i < 2; ++i) { Dt[i] = dact ? (q0[i] > thr ? q0[i] : thr) : 1.0; Dlam[i] = dact ? mu0 * rcp_nr(Dt[i]) : 0.0; }
                alpha_prev = 1.0; stp_local = 1e300;
                WSYNC();
            };
            if (!solved)
            for (; it < itmax + (cons ? fbit : 0); ++it) {
                if (it == 0) __builtin_amdgcn_s_setprio(1);
                if (it == 2) __builtin_amdgcn_s_setprio(2);
                if (it == 4) __builtin_amdgcn_s_setprio(3);
                int lz = lane;
                asm volatile("" : "+v"(lz));
                const bool uz = lz < n;
                double ru, mu, Dbar, S_i, dlam_tot, g56_tot;
                {
                    double musum = 0.0, cmax = 0.0, rineq = 0.0;
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
                    Dbar = uact ? Rj + G0 * G2 * rcp_nr(G0 + G2) + G1 * G3 * rcp_nr(G1 + G3) : 1.0;
                    if (!hres) slot_fetch<LD::NSLOT>(Hp, hsl, lane);        // H (S = 4: and Hb) back from the slot, unless the last solve prefetched it
                    cb[lane] = uact ? du : 0.0;
                    const double dlam_pref = wave_scan_incl<OpSum>(dact ? (Dlam[1] - Dlam[0]) : 0.0);     // lanes = stages
                    dlam_tot = rdlane(dlam_pref, 63);
                    sb[lane] = dlam_tot - dlam_pref;                         // suffix over stages > lane
                    const double Ssuf_incl = wave_scan_incl<OpSum>(dact ? G56 : 0.0);
                    g56_tot = rdlane(Ssuf_incl, 63);
                    sb2[lane] = g56_tot - Ssuf_incl;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // H has landed
                    WSYNC();
                    double hdu = 0.0;
                    {
                        double Rd3[3];
#pragma unroll
                        for (int m = 0; m < 3; ++m) Rd3[m] = cb[16 * m + (lane & 15)];
                        double hv[n];
                        sym_row_40_b<NR>(hv, row_addr(Hp, Hb, lz), lds_byte_addr(Hp + (uz ? lz : 0)));
                        static_for<0, n>([&](auto cc) __attribute__((always_inline)) {
                            constexpr int c = decltype(cc)::value;
                            fmac_rowbc_ld<c % 16>(hdu, Rd3[c / 16], hv[c]);
                        });
                    }
                    hdu += zgrad(lz);
                    // stationarity of the inputs / of z_s with the dynamics multipliers of the cuts left out (the primal Newton step does
                    // not depend on them: rowqp_core.h); the Bbar lanes carry the linear coupling row, whose residual is zero
                    ru = uact ? hdu + Rj * du + PK_G0 - lam[0] + lam[1] + (ji ? h * sb[ki] : 0.0)
                              : (zact ? hdu + PK_G0 + (lane == n + 6 ? dlam_tot : 0.0) : 0.0);
                    S_i = h * h * sb2[uact ? ki : 0];
                    {
                        double rnan;
                        const double rd0 = du + sl - PK_DL - t[0], rd1 = -du + su + PK_DUU - t[1], rd2 = sl - t[2], rd3 = su - t[3];
                        const double rsl = rho_l - lam[0] - lam[2], rsu = rho_u - lam[1] - lam[3];
                        const double Drd0 = dx6 - PK_DDL - Dt[0], Drd1 = PK_DDU - dx6 - Dt[1];
                        double ra = OpMaxNan::f(fabs(rsl), fabs(rsu));
                        ra = OpMaxNan::f(ra, fabs(rd0)); ra = OpMaxNan::f(ra, fabs(rd1));
                        const double rb = OpMaxNan::f(fabs(Drd0), fabs(Drd1));
                        rineq = OpMaxNan::f(uact ? ra : 0.0, dact ? rb : 0.0);
                        rnan = OpMaxNan::f(fabs(ru), uact ? OpMaxNan::f(fabs(rd2), fabs(rd3)) : 0.0);      // anything non-finite anywhere is a failure:
                        rineq = OpMaxNan::f(rineq, rnan * 0.0);                                             // 0 if finite, NaN otherwise -- rides in the same reduction
                    }
                    // start value of the tracked stationarity residual (only at a start point: rstat < 0)
                    double rs0 = 0.0;
                    if (rstat < 0.0) {
                        if (warmed) {
                            rs0 = OpMaxNan::f(uact ? fabs(lam[1] - lam[0]) : 0.0, dact ? fabs(Dlam[1] - Dlam[0]) : 0.0);
                        } else {
                            const double r_i = Rj * (ubar_i - yrs[(sc >> 1) * 9 + 7 + (sc & 1)]);
                            const int ks = lane < N ? lane : 0;
                            const double g6 = Ts * cf->W[6] * (dx6 + xbs[ks * 7 + 6] - yrs[ks * 9 + 6]);
                            rs0 = OpMaxNan::f(uact ? fabs(r_i - lam[0] + lam[1]) : 0.0, dact ? fabs(g6 - Dlam[0] + Dlam[1]) : 0.0);
                            rs0 = OpMaxNan::f(rs0, rsx);
                        }
                    }
                    const double mus_w = wave_reduce<OpSum>(musum);
                    const double cmx_w = wave_reduce<OpMax0>(cmax);
                    const double rin_w = wave_reduce<OpMaxNan>(rineq);
                    const double stp_w = wave_reduce<OpMax0>(stp_local);
                    double rs0_w = 0.0;
                    if (rstat < 0.0) rs0_w = wave_reduce<OpMaxNan>(rs0);
                    if (lane == 0) {
                        RED(wv_, 0, 0) = mus_w; RED(wv_, 0, 1) = cmx_w; RED(wv_, 0, 2) = rin_w; RED(wv_, 0, 3) = stp_w; RED(wv_, 0, 4) = rs0_w;
                    }
                    XSYNC();
                    double msum = 0.0, cmx = 0.0, rin = 0.0, stp = -INFINITY, rs0a = 0.0;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        msum += RED(s, 0, 0); cmx = fmax(cmx, RED(s, 0, 1)); rin = OpMaxNan::f(rin, RED(s, 0, 2));
                        stp = fmax(stp, RED(s, 0, 3)); rs0a = OpMaxNan::f(rs0a, RED(s, 0, 4));
                    }
                    mu = uni(msum * inv_nineq); step = uni(stp); cmx = uni(cmx); rin = uni(rin);
                    if (rstat < 0.0) rstat = uni(rs0a);
                    const double rmax = OpMaxNan::f(rin, rstat);
                    if (!(mu == mu) || !(rmax == rmax)) { failed = true; break; }
                    if (cmx <= tol_comp && step <= tol_step &&
                        (rmax <= tol_res || (it > 0 && rmax > 0.1 * rmax_prev && rmax <= ADMPC_IPM_FLOOR_CAP * tol_res))) break;      // admpc.h: stopping test
                    rmax_prev = rmax;
                }
                SEG_STAMP(4);
                if (fbit > 0 && !cons && it >= fbit) {
                    cons = true; warmed = false;
                    cold_start();
                    rmax_prev = 0.0; rstat = -1.0;
                    --it;
                    continue;
                }
                factorise(Dbar, (uz && ji) ? S_i : 0.0, lz, !first, h);
                hres = false;
                schur(lz);
                if (lz == 0) IFm[LD::IF_G56] = g56_tot;
                SEG_STAMP(5);
                XSYNC(); if (wv_ == 0) { if constexpr (S == 2) interface_factor2(ifb, YM, lz); else interface_factor<S>(ifb, lz); SEG_STAMP(13); }
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
                bool restarted = false;
#pragma unroll 1
                for (int ps = 0; ps < 2; ++ps) {
                    const double c0 = rc[0] * it_[0], c1 = rc[1] * it_[1], c2 = rc[2] * it_[2], c3 = rc[3] * it_[3];
                    const double e1 = rsl + c0 + c2 + G0 * rd0 + G2 * rd2;
                    const double e2 = rsu + c1 + c3 + G1 * rd1 + G3 * rd3;
                    const double etal = c0 + G0 * rd0 - G0 * e1 * iG02;
                    const double etau = -c1 - G1 * rd1 + G1 * e2 * iG13;
                    const double ek = dact ? (Drc[0] * Dit[0] + G5 * Drd0) - (Drc[1] * Dit[1] + G6 * Drd1) : 0.0;
                    const double epref = wave_scan_incl<OpSum>(ek);
                    const double ek_tot = rdlane(epref, 63);
                    sb[lane] = ek_tot - epref;
                    WSYNC();
                    const double y = uact ? -(ru + etal + etau + (ji ? h * sb[ki] : 0.0)) : (zact ? -(ru + (lane == n + 6 ? ek_tot : 0.0)) : 0.0);
                    const double x = coupled_solve(y, lz);
                    if (ps == 1) { slot_fetch<LD::NSLOT>(Hp, hsl, lz); hres = true; }      // the factor is dead: next iteration's H under the step-length work
                    ddu = uact ? x : 0.0;
                    cb[lane] = ddu;
                    WSYNC();
                    const double dz6 = first ? 0.0 : uni(IFm[LD::IF_DZ + 6]);
                    const double du1_stage = lane < N ? cb[2 * lane + 1] : 0.0;
                    const double pre = wave_scan_incl<OpSum>(du1_stage);
                    const double ddx6 = dz6 + h * (pre - du1_stage);
                    dsl = -(e1 + G0 * ddu) * iG02;
                    dsu = -(e2 - G1 * ddu) * iG13;
                    dt[0] = ddu + dsl + rd0; dt[1] = -ddu + dsu + rd1; dt[2] = dsl + rd2; dt[3] = dsu + rd3;
                    const double Gs[4] = { G0, G1, G2, G3 };
                    double rr = 0.0, sdd = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dlam[i] = -rc[i] * it_[i] - Gs[i] * dt[i];
                        rr = fmax(rr, uact ? fmax(-dt[i] * it_[i], -dlam[i] * il_[i]) : 0.0);
                        sdd += uact ? dt[i] * dlam[i] : 0.0;
                    }
                    Ddt[0] = ddx6 + Drd0;  Ddlam[0] = -Drc[0] * Dit[0] - G5 * Ddt[0];
                    Ddt[1] = -ddx6 + Drd1; Ddlam[1] = -Drc[1] * Dit[1] - G6 * Ddt[1];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        rr = fmax(rr, dact ? fmax(-Ddt[i] * Dit[i], -Ddlam[i] * Dil[i]) : 0.0);
                        sdd += dact ? Ddt[i] * Ddlam[i] : 0.0;
                    }
                    const double rr_w = wave_reduce<OpMax0>(rr);
                    const double sdd_w = wave_reduce<OpSum>(sdd);
                    if (lane == 0) { RED(wv_, 2 + ps, 0) = rr_w; RED(wv_, 2 + ps, 1) = sdd_w; }
                    XSYNC();
                    double rra = 0.0, sdda = 0.0;
#pragma unroll
                    for (int s = 0; s < S; ++s) { rra = fmax(rra, RED(s, 2 + ps, 0)); sdda += RED(s, 2 + ps, 1); }
                    rra = uni(rra); sdda = uni(sdda);
                    const double amax = rra > 1.0 ? rcp_nr(rra) : 1.0;
                    if (ps == 0) {
                        // complementarity after the affine step: sum (t + a dt)(lam + a dlam) = (1 - a) sum t lam + a^2 sum dt dlam (the
                        // predictor's right-hand side is t lam, so t dlam + lam dt = -t lam exactly: rowqp_core.h uses the same identity)
                        mu_aff = ((1.0 - amax) * mu * (double)(8 * NT + 2 * (NT - 1)) + amax * amax * sdda) * inv_nineq;
                        double sigma = mu_aff * rcp_nr(mu); sigma = sigma * sigma * sigma;
                        if (alpha_prev < ADMPC_IPM_BLOCKED_STEP) sigma = 1.0;
                        const double smu = fmax(sigma * mu, ADMPC_IPM_MU_FLOOR * tol_comp);
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
                            warmed = false;
                            cold_start();
                            restarted = true;
                        } else {
                        alpha_prev = alpha;
                        rstat = (1.0 - alpha) * rstat;
                        stp_local = uact ? fabs(alpha * ddu) : 0.0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { t[i] = fmax(t[i] + alpha * dt[i], IPM_FLOOR); lam[i] = fmax(lam[i] + alpha * dlam[i], IPM_FLOOR); }
                        du += alpha * ddu; sl += alpha * dsl; su += alpha * dsu;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            Dt[i] = dact ? fmax(Dt[i] + alpha * Ddt[i], IPM_FLOOR) : 1.0;
                            Dlam[i] = dact ? fmax(Dlam[i] + alpha * Ddlam[i], IPM_FLOOR) : 1.0;
                        }
                        dx6 += dact ? alpha * ddx6 : 0.0;
                        if (!first && lane < 7) IFm[LD::IF_Z + lane] += alpha * IFm[LD::IF_DZ + lane];
                        }
                    }
                    WSYNC();
                    SEG_STAMP(9);
                }
                if (restarted) { rmax_prev = 0.0; rstat = -1.0; }
            }
        }
        if (failed) break;
        // ---- the cut states of the returned inputs: Bbar_s U_s + c_s per segment (Hb is in place: S = 4 brought it back with H behind the last solve), chained below
        if (S > 1 && !last) {
            LAUNDER_LANE(lane);
            const bool uact = lane < n;
            const double duv = uact ? du : 0.0;
            const int li = uact ? lane : 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (S = 4: Hb came back with the last prefetch)
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const double v = wave_reduce<OpSum>(Hb[(bslot + r) * BS + li] * duv);
                if (lane == 0) IFm[LD::IF_BU + r] = v + IFm[LD::IF_C + r];
            }
        }
        XSYNC();
        SEG_STAMP(10);
        }   // pass
        { LAUNDER_LANE(lw); if (wv_ == 0 && lw == 0 && itersg) itersg[inst] = it; }
        if (failed) {
            // non-finite QP data: acados returns before the update -- the iterate stays as it is, status 4, cost +inf
            { LAUNDER_LANE(lw); if (wv_ == 0 && lw == 0) { statusg[inst] = ADMPC_STATUS_QP_FAILURE; if (costg) costg[inst] = INFINITY; } }
            XSYNC();
            __builtin_amdgcn_s_setprio(0);
            continue;
        }

        // =================================================================================================================
        // phase E (H6): expansion of the segment's states from its cut state, full step, cost, status
        // =================================================================================================================
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const int ji = lane & 1;
            const bool uact = lane < n;
            const int sc = uact ? lane : 0;
            const double Ts = cf->Ts, h = cf->Ts;
            const double Rj = Ts * cf->W[NX + ji];
            const double rho_l = Ts * cf->zl, rho_u = Ts * cf->zu;
            const int r6 = lane < 6 ? lane : 0;
            const int r7 = lane < NX ? lane : 0;
            const double wq = lane < NX ? Ts * cf->W[r7] : 0.0, wqe = lane < NX ? cf->We[r7] : 0.0;
            // z_1 = (Bbar U + c)_0, z_{t+1} = Abar_t z_t + (Bbar U + c)_t: lanes 0..6 = components
            double zs = 0.0;
            if (!first) {
                zs = ifb[LD::IF_BU + r7];
#pragma unroll 1
                for (int t_ = 1; t_ < wv_; ++t_) {
                    cb[lane] = zs;
                    WSYNC();
                    const double* Ft = ifb + t_ * LD::IFS;
                    double a = Ft[LD::IF_BU + r7];
#pragma unroll
                    for (int c = 0; c < 7; ++c) a = fma(Ft[LD::IF_ABAR + r7 * 8 + c], cb[c], a);
                    WSYNC();
                    zs = a;
                }
            }
            WSYNC();
            stage_dq_seg(dqE, xbs, yrs, yre, lane);
            du = uact ? du : 0.0;
            dus[lane] = du;
            const double ubar_i = ubs[sc];
            const double uref_i = yrs[(sc >> 1) * 9 + 7 + (sc & 1)];
            double dx = lane < NX ? (first ? x0g[(size_t)inst * NX + r7] - xbg[r7] : zs) : 0.0;
            WSYNC();
            bool bad = false;
            double J = 0.0;
            static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const double e = dx + dqE[k * 7 + r7];
                if (k < N || last) J += 0.5 * (k < N ? wq : wqe) * e * e;      // stage k0 + 20 is the next segment's
                if (!(fabs(dx) <= 1e300)) bad = true;
                if (lane < NX) dqE[k * 7 + lane] = dx;
                if constexpr (k < N) {
                    const double* Gk = GT + k * GTS;
                    const double u0 = dus[2 * k], u1 = dus[2 * k + 1];
                    // every lane loads rows 0..5 (lanes >= 6 read row 0 and drop the result), row 6 = [e6, 0, h] in lane 6: see admpc_fused20.hip
                    const double bk = bl[k * 7 + r7];
                    double acc = bk + (lane < 2 ? dx : 0.0);
                    double gg[5];
#pragma unroll
                    for (int c = 0; c < 5; ++c) gg[c] = Gk[c * 6 + r6];
                    const double b0 = Gk[5 * 6 + r6], b1 = Gk[6 * 6 + r6];
                    acc += b0 * u0 + b1 * u1;
                    const double acc6 = fma(h, u1, bk + dx);
                    fmac_rowbc<2>(acc, dx, gg[0]); fmac_rowbc<3>(acc, dx, gg[1]); fmac_rowbc<4>(acc, dx, gg[2]);
                    fmac_rowbc<5>(acc, dx, gg[3]); fmac_rowbc<6>(acc, dx, gg[4]);
                    dx = lane < 6 ? acc : (lane == 6 ? acc6 : 0.0);
                }
            });
            const double unew = ubar_i + du;
            if (uact && !(fabs(unew) <= 1e300)) bad = true;
            double Ju = 0.0;
            if (uact) {
                const double e = unew - uref_i;
                Ju = 0.5 * Rj * e * e;
                if (unew < cf->lbu[ji]) Ju += rho_l * (cf->lbu[ji] - unew);
                if (unew > cf->ubu[ji]) Ju += rho_u * (unew - cf->ubu[ji]);
            }
            const double Jt = wave_reduce<OpSum>(J + Ju);
            const bool anyb = __any(bad);
            if (lane == 0) { RED(wv_, 1, 0) = Jt; RED(wv_, 1, 1) = anyb ? 1.0 : 0.0; }
            XSYNC();                                        // every wave has read what it needs of xbar (the rows at the cuts are shared)
            double Jall = 0.0; bool badall = false;
#pragma unroll
            for (int s = 0; s < S; ++s) { Jall += RED(s, 1, 0); badall = badall || RED(s, 1, 1) != 0.0; }
            const int status = badall ? ADMPC_STATUS_QP_FAILURE : ADMPC_STATUS_SUCCESS;
            if (status == 0) {
                // rows k0 + 1 .. k0 + 20 (the first segment: row 0 as well); the row at the cut is written by the segment in front of it
                const int lo = first ? 0 : NX;
#pragma unroll
                for (int i0 = 0; i0 < (N + 1) * NX; i0 += WAVE) { const int i = i0 + lane; if (i >= lo && i < (N + 1) * NX) STG(xbs + i, LDG(xbs + i) + dqE[i]); }
                if (uact) STG(ubs + lane, unew);
            }
            if (wv_ == 0 && lane == 0) {
                if (costg) costg[inst] = status == 0 ? Jall : INFINITY;
                statusg[inst] = status;
            }
            XSYNC();
        }
        __builtin_amdgcn_s_setprio(0);
        SEG_STAMP(12);
#ifdef ADMPC_PHASE_TIMERS
        if (threadIdx.x == 0 && inst < 16384) { g_seg_trace[4 * inst] = tr_t0; g_seg_trace[4 * inst + 1] = seg_real(); g_seg_trace[4 * inst + 2] = blockIdx.x; g_seg_trace[4 * inst + 3] = tr_t1; }
#endif
    }
    SEG_FLUSH();
    // ---- every workgroup has drawn exactly one ticket beyond the batch; the last one to leave clears tickets and bins for the next launch
    if (wv_ == 0) {
        LAUNDER_LANE(lane0);
        int gone = 0;
        if (lane0 == 0) gone = atomicAdd(sched + 1, 1);
        gone = __builtin_amdgcn_readfirstlane(gone);
        if (gone == (int)gridDim.x - 1) { sched[lane0] = 0; sched[64 + lane0] = 0; __threadfence(); }
    }
#undef PK_DL
#undef PK_DUU
#undef PK_G0
#undef PK_DDL
#undef PK_DDU
#undef RED
}

}  // namespace

template <int S>
static void seg_launch(int num_cu, hipStream_t st, const AdmpcConfig* d_cfg, int B, int qmask,
        const double* x0, const double* yref, const double* yref_e, const double* p, double* xbar, double* ubar,
        double* cost, int32_t* stat, int32_t* iters, int first, int* sched2, int cap, int flip, double* hslot)
{
    const int lds = SegLds<S>::total * (int)sizeof(double);
    // persistent grid: as many workgroups per CU as LDS (160 KB) and wave slots (two per SIMD) allow
    int per_cu = (160 * 1024) / lds;
    if (per_cu > 8 / S) per_cu = 8 / S;
    if (per_cu < 1) per_cu = 1;
    int grid = num_cu * per_cu; if (grid > B) grid = B;
    static bool prepared = false;
    if (!prepared) {
        (void)hipFuncSetAttribute((const void*)admpc_seg_kernel<S, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)admpc_seg_kernel<S, 127>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        prepared = true;
    }
    // two scheduler states, used alternately: the order kernel of this launch zeroes the header of the next (work_order.h)
    const size_t one = admpc_fused20_state_ints(cap);      // the layout of a scheduler state is admpc_fused20.hip's (its order kernel also resets the expansion queue behind the lists)
    int* const sched = sched2 + (flip ? one : 0);
    int* const sched_next = sched2 + (flip ? 0 : one);
    const int kcap = grid == B ? 0 : cap;      // the batch fits the grid: no work order (work_order.h: f20_next)
    if (kcap) hipLaunchKernelGGL(admpc_f20_order_kernel, dim3((B + 255) / 256), dim3(256), 0, st, d_cfg, B, x0, yref, yref_e, sched, cap, sched_next);
    if (qmask == 7)
        hipLaunchKernelGGL((admpc_seg_kernel<S, 7>), dim3(grid), dim3(WAVE * S), lds, st, d_cfg, B, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched, kcap, hslot);
    else
        hipLaunchKernelGGL((admpc_seg_kernel<S, 127>), dim3(grid), dim3(WAVE * S), lds, st, d_cfg, B, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched, kcap, hslot);
}

// ---- host side (called by solve_impl in admpc_kernels.hip)
extern "C" {

// timer builds only: read and clear the phase counters
int admpc_debug_seg_ticks(unsigned long long* out16)
{
#ifdef ADMPC_PHASE_TIMERS
    unsigned long long z[16] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_seg_ticks), sizeof z) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_seg_ticks), z, sizeof z) != hipSuccess) return -1;
    return 0;
#else
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    return 1;
#endif
}

int admpc_debug_seg_trace(unsigned long long* out, int n_inst)
{
#ifdef ADMPC_PHASE_TIMERS
    if (n_inst > 16384) n_inst = 16384;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_seg_trace), (size_t)n_inst * 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return 0;
#else
    (void)out; (void)n_inst;
    return 1;
#endif
}

// bring-up builds only: copy the dump buffer (4 waves x 2 points x 16384 doubles) to the host; 1 when the build carries none
int admpc_debug_seg(double* out, int inst)
{
#ifdef SEG_DEBUG
    if (out) {
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_seg_dbg), sizeof(double) * 4 * 2 * SEG_DBG_W) != hipSuccess) return -1;
    } else if (hipMemcpyToSymbol(HIP_SYMBOL(g_seg_dbg_inst), &inst, sizeof(int)) != hipSuccess) return -1;
    return 0;
#else
    (void)out; (void)inst;
    return 1;
#endif
}

// doubles of the slot buffer: one packed H per resident wave (at most 8 waves per CU)
__attribute__((visibility("hidden"))) // (at most eight waves per CU; S > 2 parks its 14 constant border rows behind H: SegLds::NSLOT)
size_t admpc_seg_slot_doubles(int num_cu) { return (size_t)num_cu * 8 * SegLds<4>::NSLOT; }

// horizons this unit serves (fp64): N = 20 S, S = 2, 3, 4
__attribute__((visibility("hidden"))) int admpc_seg_supports(int N) { return N == 40 || N == 60 || N == 80; }

__attribute__((visibility("hidden"))) int admpc_seg_lds_bytes(int N)
{
    switch (N / 20) {
        case 2: return SegLds<2>::total * (int)sizeof(double);
        case 3: return SegLds<3>::total * (int)sizeof(double);
        default: return SegLds<4>::total * (int)sizeof(double);
    }
}

// grid: persistent workgroups of S waves; sched2: admpc_fused20_sched_ints(cap) ints = two scheduler states, flip selects this launch's (admpc_fused20.hip)
__attribute__((visibility("hidden"))) void admpc_seg_launch(int N, int num_cu, hipStream_t st, const AdmpcConfig* d_cfg, int B, int qmask,
        const double* x0, const double* yref, const double* yref_e, const double* p, double* xbar, double* ubar,
        double* cost, int32_t* stat, int32_t* iters, int first, int* sched2, int cap, int flip, double* hslot)
{
    switch (N / 20) {
#ifndef SEG_DEV_ONLY_S4
        case 2: seg_launch<2>(num_cu, st, d_cfg, B, qmask, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched2, cap, flip, hslot); break;
#endif
#if defined(SEG_DEV_ONLY_S4)
        default: seg_launch<4>(num_cu, st, d_cfg, B, qmask, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched2, cap, flip, hslot); break;
#elif !defined(SEG_DEV_ONLY_S2)      // development builds: one instantiation compiles in a third of the time
        case 3: seg_launch<3>(num_cu, st, d_cfg, B, qmask, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched2, cap, flip, hslot); break;
        default: seg_launch<4>(num_cu, st, d_cfg, B, qmask, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched2, cap, flip, hslot); break;
#else
        default: break;
#endif
    }
}

}
