// admpc_fused20.hip -- the N = 20 fp64 SQP-RTI step (BASELINE configs[1..3]) as ONE persistent kernel for gfx950.
//
// One wavefront (= one workgroup of 64 lanes) owns one MPC instance from its inputs to its outputs and then draws the next one
// from a ticket counter.  Nothing but the algorithmic inputs and outputs of SURVEY 8a crosses HBM: the condensed Hessian (6.6 KB),
// its factor (6.6 KB) and every intermediate live in the wave's 18.2 KB of LDS or in registers -- eight instances per CU, two waves
// per SIMD; the packed linearisation (7.8 KB) aliases that space and is recomputed in front of the expansion (or parked in a per-wave
// slot of global memory when the model carries GP residuals).  Phases of an instance (reference = data_driven_mpc/ros_gp_mpc/src/ad_mpc/...):
//   A  H0/H1  ERK4 + forward sensitivities of all 20 stages           ad_3d_optimizer.py:280-310, acados ERK
//             (A1: lanes (stage, third) integrate the state and table      (acados_solver_sim_car.c:655-665)
//              the Jacobian entries of the four RK stages in LDS; A2: the same lanes integrate their 2-3 sensitivity columns
//              from the tables -- the split keeps the phase inside the 256-register budget of two waves per SIMD)
//   C  H2-H4  Gauss-Newton cost, bounds, full condensing                ad_3d_optimizer.py:146-199; acados_solver_sim_car.c:145
//   D  H5     unconstrained trial, Mehrotra predictor-corrector on the dense 40-input QP (reference: HPIPM, :688-692)
//   E  H6     state expansion, full step, cost, status                  acados_solver_sim_car.c:647-648,677
// The phase bodies descend from the four-kernel pipeline of rounds 1-2 (kernels A, C, D, E in admpc_kernels.hip, `make legacy` only;
// DESIGN section 4); what changed is where the data lives and that no instance waits for a kernel boundary: the slowest instance
// of a batch starts at once instead of after everybody's linearisation and condensing.
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
#ifndef F20_NT
#define F20_NT 0
#endif
// 1 (shipped): the Hessian H = sum_k Gamma_k' Q Gamma_k is accumulated on the matrix pipe (v_mfma_f64_16x16x4_f64 tiles), the stage
// recursion stays on the vector pipe; 0: everything by v_fmac_f64_dpp rows on the vector pipe; 2: the whole condensing -- recursion
// Gamma_{k+1} = A_k Gamma_k, free response, Hessian, reduced gradient -- in MFMA tiles (correct, 106 GPU tests green, but 1 % SLOWER
// than 1: a 16 x 16 x 4 tile carries 7 x 7 useful products in the recursion and the dependent chain of 42 of them is latency-bound) (A/B: scripts/probes/mfma_condense_probe.hip, profiles/r3/mfma_condense_ab.txt)
#ifndef F20_MFMA
#define F20_MFMA 1
#endif
// 1 (shipped): every LDS read of a condensing stage is issued in front of the stage (458 -> 97 s_waitcnt in the phase; an instruction of
// any kind costs its wave an issue slot, and the single-wave time is what the end of a launch runs at); 0: where hipcc puts them
#ifndef F20_CPREF
#define F20_CPREF 1
#endif
// 1: the free response xhat_k of the condensing rides in lane 40 as a 41st column of Gamma (b_k enters it through a per-lane LDS
// address, the other lanes read the components they need from the exchange buffer every lane writes anyway) instead of being propagated
// redundantly by all 64 lanes: 30 multiply-adds per stage.  Needs F20_CPREF and F20_MFMA == 1.  The same bits.
#ifndef F20_XHLANE
#define F20_XHLANE 1
#endif
// 1: the 30 entries of A_k reach the propagation's multiply-adds through DPP row broadcasts -- every 16-lane row holds them in TWO registers
// (lane e: entries e and 16 + e, two ds_read_b64 per stage) -- instead of fifteen ds_read_b128 that hand all 64 lanes the same 16 bytes:
// 15 KB of LDS return traffic per stage for 240 bytes of information.  The LDS pipe of a CU is shared by its eight waves and was 76 % busy
// over the whole launch (SQ_ACTIVE_INST_LDS).  Needs F20_XHLANE (no uniform operands left in the phase).  The same bits.
#ifndef F20_ADPP
#define F20_ADPP 1
#endif
#if F20_ADPP && !F20_XHLANE
#error "F20_ADPP needs F20_XHLANE"
#endif
#if F20_XHLANE && !(F20_CPREF && F20_MFMA == 1)
#error "F20_XHLANE needs F20_CPREF and F20_MFMA == 1"
#endif
#ifndef F20_TICKET_AHEAD
#define F20_TICKET_AHEAD 1
#endif
// Deferred expansions (0, shipped: off; 1: the instances drawn by ticket push their expansion -- phase A once more and phase E -- into
// queues that the waves pop when the ticket queue is dry: jobs of 15 us instead of 54 at the end of a launch).  Built, bit-identical
// (scripts/defer_check.py), and measured SLOWER: 19.2 M against 20.9 M solves/s at configs[1], 22.1 against 24.1 M at B = 8192, two batches
// in flight 20.6 against 23.1 M -- an expansion that changes waves pays ~6 us of dependent trips to the memory side (queue scan, pop,
// entry, du; the instance's inputs miss the other XCD's L2) on a 15 us job, more than the shorter ramp returns.  With ONE queue the
// counter line was the bottleneck (same-line atomics retire at ~10 ns: 35 % slower), as compare-and-swap pops 35 ms per step.  See
// `expansion queue` in the kernel; kept as an A/B switch (make variant EXTRA=-DF20_DEFER=1).
#ifndef F20_DEFER
#define F20_DEFER 0
#endif
#if F20_DEFER && !F20_TICKET_AHEAD
#error "F20_DEFER needs F20_TICKET_AHEAD"
#endif
#ifndef F20_TOKTRAP
#define F20_TOKTRAP 0
#endif

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- optional per-phase wave-time accounting (build with -DADMPC_PHASE_TIMERS: `make timers`): s_memtime ticks (100 MHz) summed
//      over all waves: 0 ticket draw, 1 A1 (state RK4 + model), 2 A2 (sensitivity columns), 3 C (condensing), 4 D trial,
//      5 D interior-point iterations, 6 E (expansion + outputs); [8] wave-time from kernel start to the wave's exit
#if defined(ADMPC_PHASE_TIMERS) || defined(ADMPC_F20_TRACE)
__device__ unsigned long long g_f20_trace[4 * 8192];      // per instance (first 8192): start, end (s_memrealtime, 100 MHz), block, IPM start
__device__ __forceinline__ unsigned long long f20_real() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
#define F20_TRACE_BEGIN() const unsigned long long tr_t0 = f20_real(); unsigned long long tr_t1 = 0
#define F20_TRACE_MID() tr_t1 = f20_real()
#define F20_TRACE_END(inst) do { if (threadIdx.x == 0 && (inst) < 8192) { g_f20_trace[4 * (inst)] = tr_t0; g_f20_trace[4 * (inst) + 1] = f20_real(); g_f20_trace[4 * (inst) + 2] = blockIdx.x; g_f20_trace[4 * (inst) + 3] = tr_t1; } } while (0)
#else
#define F20_TRACE_BEGIN() do { } while (0)
#define F20_TRACE_MID() do { } while (0)
#define F20_TRACE_END(inst) do { } while (0)
#endif
#ifdef ADMPC_PHASE_TIMERS
__device__ unsigned long long g_f20_ticks[16];
__device__ __forceinline__ unsigned long long f20_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
#define F20_DECL() unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long ph_t0 = f20_now(); unsigned long long ph_last = ph_t0
#define F20_STAMP(k) do { const unsigned long long t_ = f20_now(); ph_acc[k] += t_ - ph_last; ph_last = t_; } while (0)
#define F20_FLUSH() do { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&g_f20_ticks[q_], ph_acc[q_]); atomicAdd(&g_f20_ticks[8], f20_now() - ph_t0); atomicMax(&g_f20_ticks[9], f20_now() - ph_t0); } } while (0)
#elif defined(F20_MARKS)      // listing with phase markers for scripts/asm_phase_stats.py (hipcc -S -DF20_MARKS): the code after stamp k is phase k + 1
#define F20_DECL() do { } while (0)
#define F20_STAMP(k) asm volatile("; MARK_after" #k)
#define F20_FLUSH() do { } while (0)
#else
#define F20_DECL() do { } while (0)
#define F20_STAMP(k) do { } while (0)
#define F20_FLUSH() do { } while (0)
#endif

#include "model_dev.h"
#include "dense40.h"
#include "cond_common.h"

// Inputs and outputs of an instance are touched once (xbar three times, minutes of L2 time apart): marked non-temporal so that
// they do not push the waves' slot buffers out of the L2 (measured: no effect on traffic, 0.5 % slower: off)
#if F20_NT
#define LDG(p) __builtin_nontemporal_load(p)
#define STG(p, v) __builtin_nontemporal_store(v, p)
#else
#define LDG(p) (*(p))
#define STG(p, v) (*(p) = (v))
#endif
// stage_dq of cond_common.h with the loads above
template <int NN>
__device__ __forceinline__ void stage_dq_nt(double* __restrict__ dq, const double* __restrict__ xb, const double* __restrict__ yr,
                                            const double* __restrict__ yre, const int lane) {
    constexpr int CNT = (NN + 1) * NX, IT = (CNT + WAVE - 1) / WAVE;
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

// LDS map of one instance (doubles): what the interior point needs and nothing else -- 18.2 KB, eight instances per CU (two waves per
// SIMD).  The exchange buffers keep the relative layout dense40.h's col_head assumes (sb = cb + 128).  The other phases alias it:
//   A   JT [0, 1960) Jacobian tables of the RK stages, bl [1960, 2100) defects; then GT [0, 840) (written when the tables are dead)
//   C   reads GT, bl; dq [860, 1008), gam [1008, 1456); leaves H in [0, 820) (row store after the last read of GT)
//   E   GT, bl once more in the places of phase A (second run of phase A, or read back from the slot buffer); dq, du in C's places
// The packed linearisation (GT, bl: 7.8 KB) is needed again behind the interior point, by the expansion, and LDS cannot keep it at this
// occupancy: see `park_gt` in the kernel for the two ways across.
struct FusedLds {
    static constexpr int N = 20, NTRI = 820;
    static constexpr int oH = 0, oL = oH + NTRI, oPark = oL + NTRI, oCb = oPark + 5 * 64;
    static constexpr int oSch = oCb + 4 * 64;                           // [2][64] ints: inclusive scan of the bin counts (most expensive bin first), the counts
    static constexpr int total = oSch + 64;                             // 2280 doubles = 18 240 B
    static constexpr int JTS = 24, JTK = 4 * JTS + 2;                   // Jacobian entries per (stage, RK stage); doubles per stage: 98, not 96 --
                                                                        // 768 B apart the 20 stages of a table access all hit one bank group
    static constexpr int oJT = 0, oBlA = N * JTK, oGTC = 0, oDqC = 860, oGam = oDqC + 148;
    static constexpr int SLOT = N * GTS + N * NX;                       // doubles per wave in the slot buffer
    static_assert(oBlA + N * NX <= total && oGTC + N * GTS <= oDqC && oGam + (NX + 1) * 64 <= oPark, "LDS aliases");
};

// ---- work order.  A wave owns an instance for 40 us (the trial solves it) up to 200 us (13 interior-point iterations), and a batch
// of 4096 is two rounds of the 2048 resident waves: an expensive instance that is drawn late IS the kernel's run time.  A pre-pass
// (one thread per instance, a few loads) bins the instances by a kinematic estimate of how far the longitudinal input has to leave
// its box: the constant acceleration that carries the vehicle from x0 to the along-track position of the terminal reference,
// a = 2 (s_ref - v_x T) / T^2, against [lbu_0, ubu_0].  On the config-2 scenarios every instance that needs 8 or more iterations
// is in the first round with it, and 1837 of the 1841 that need the interior point at all (correlation with the iteration count
// 0.80).  A heuristic: it orders work and nothing else -- results do not depend on the draw order.
//   sched: [0] ticket counter, [1] exit counter, [F20_BINS0 + q] instances in bin q, [F20_HDR + q * cap + j] j-th instance of bin q
// wave priority by interior-point iteration (measured: 0/2/4 -> 0.250 ms per step, 1/3/6 0.254, 3/6 only 0.255, none 0.269)
#ifndef F20_PRIO_IT1
#define F20_PRIO_IT0 0
#define F20_PRIO_IT1 2
#define F20_PRIO_IT2 4
#endif
#ifndef F20_CLASS_PRIO
#define F20_CLASS_PRIO 0
#endif
#ifndef F20_PAIR_REV
#define F20_PAIR_REV 0
#endif
#include "work_order.h"

static_assert(WAVE == 64, "one wavefront per workgroup: WSYNC is a wave-level fence in this build");
template <int QMASK>
__global__ __launch_bounds__(WAVE, 2) void admpc_fused20_kernel(const AdmpcConfig* __restrict__ cfg, int B,
                                                                const double* __restrict__ x0g, const double* __restrict__ yrefg,
                                                                const double* __restrict__ yrefeg, const double* __restrict__ pg,
                                                                double* __restrict__ xbarg, double* __restrict__ ubarg,
                                                                double* __restrict__ costg, int32_t* __restrict__ statusg,
                                                                int32_t* __restrict__ itersg, int first_pass, int* __restrict__ sched, int cap,
                                                                double* __restrict__ slotbuf)
{
    constexpr int N = 20, n = 40;
    extern __shared__ double lds_raw[];
    double* const Hp = lds_raw + FusedLds::oH;          // packed lower-triangular rows of H
    double* const Lp = lds_raw + FusedLds::oL;          // packed strictly-lower rows of the unit factor L (M = L D L')
    double* const park = lds_raw + FusedLds::oPark;     // [5][64] per-lane constants (registers are the scarce resource)
    double* const cb = lds_raw + FusedLds::oCb;         // [64] step broadcast buffer
    double* const invd = cb + 64;                       // [64] 1 / D_jj
    double* const sb = invd + 64;                       // [64] per-stage exchange
    double* const sb2 = sb + 64;                        // [64]
    double* const JT = lds_raw + FusedLds::oJT;         // phase A: [N][4][24] Jacobian entries of the RK stages
    double* const GT = lds_raw + FusedLds::oGTC;        // phases A (end), C: packed linearisation of the instance
    double* const bl = lds_raw + FusedLds::oBlA;        // phases A, C: defects b_k
    double* const dqC = lds_raw + FusedLds::oDqC;       // phase C: xbar_k - xref_k
    double* const gam = lds_raw + FusedLds::oGam;       // phase C: [NX][64] Gamma components of the current stage
    double* const GTe = GT;                             // phase E: the linearisation again, in the same places (second run of phase A,
    double* const ble = bl;                             //          or read back from the wave's slot buffer)
    double* const dqE = lds_raw + FusedLds::oDqC;       // phase E: xbar_k - xref_k, overwritten by dx_k
    double* const dus = lds_raw + FusedLds::oGam;       // phase E: [64] du per input
    // How the linearisation gets across the interior point (LDS cannot keep it at eight instances per CU):
    //   slotbuf == NULL  phase A runs a second time in front of phase E: nothing but the instance's inputs and outputs crosses HBM
    //                    (24 MB per 4096-instance step, 1.3 x the algorithmic bytes; + 2 % time);
    //   slotbuf != NULL  it is parked in a slot buffer that belongs to the WAVE (grid x 980 doubles, reused for every instance the wave
    //                    draws, written behind phase A and read back in front of phase E: 71 MB per step).  The host chooses this when
    //                    GP residuals are configured: their kernel sums make phase A several times as expensive.
    const bool park_gt = slotbuf != nullptr;
    double* const slot = slotbuf + (size_t)blockIdx.x * FusedLds::SLOT;
#define PK_DL   park[0 * 64 + lane]
#define PK_DUU  park[1 * 64 + lane]
#define PK_G0   park[2 * 64 + lane]
#define PK_DDL  park[3 * 64 + lane]
#define PK_DDU  park[4 * 64 + lane]

    // Every phase derives its per-lane quantities from a freshly laundered lane id and reads its constants through a freshly
    // laundered config pointer: what is loop-invariant across instances must be recomputed in place -- hoisted out of the persistent
    // loop it was parked in scratch (257 SGPR lanes and 89 VGPRs in the first build) and reloaded inside the stage loops.
// lane id from v_mbcnt (a workgroup is one wave), never from threadIdx.x: v0 would stay live (and be spilled) across the whole kernel
#define LAUNDER_LANE(v) int v = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(v))
// an opaque zero offset, not an opaque pointer: the compiler keeps knowing that the config is uniform, read-only global memory (s_load)
#define LAUNDER_CFG(c) int c##_z = 0; asm volatile("" : "+s"(c##_z)); const AdmpcConfig* __restrict__ c = cfg + c##_z

    // Factorisation of the Newton matrix M = H + diag(dbar) + (s_odd on the odd columns of the u1 rows) into L D L' (LDS: Lp, invd);
    // text of kernel D (admpc_kernels.hip), see there and dense40.h for the look-ahead scheme.
    auto factorise = [&](const double dbar_, const double sodd_, const int lz_) __attribute__((always_inline)) {
        const int trz_ = lz_ * (lz_ + 1) / 2;
        const bool uz_ = lz_ < n;
        double a[n];
        newton_row_40(a, lds_byte_addr(Hp + (uz_ ? trz_ : 0)), dbar_, sodd_);
        const unsigned lrow = lds_byte_addr(Lp + (uz_ ? trz_ : 0));
        const unsigned pub_wr = lds_byte_addr(cb + lz_), pub_rd = lds_byte_addr(cb + (lz_ & 15));
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
        double Rb[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}, nlb[2] = {0.0, 0.0};
        cb[lz_] = a[0];
#pragma unroll
        for (int m = 0; m < 3; ++m) Rb[0][m] = cb[16 * m + (lz_ & 15)];
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
        const bool uz_ = lz_ < n;
        const unsigned pub = lds_byte_addr(cb + (lz_ & 15));                              // cb is free while a system is being solved
        fwd_subst_40(y, lds_byte_addr(Lp + (uz_ ? lz_ * (lz_ + 1) / 2 : 0)), pub);      // idle lanes never take part (EXEC masks)
        double x = y * invd[uz_ ? lz_ : 0];
        bwd_subst_40(x, lds_byte_addr(Lp + (uz_ ? lz_ : 0)), pub);
        return x;
    };

    // ---------------- persistent loop: first ticket = block index, later ones from one global counter ----------------
    F20_DECL();
    bool first_ticket = true;
    // Tickets.  Between two instances f20_next is three dependent trips to the L2 (the ticket counter, the bin counts, the bin's list) in
    // front of the instance's own loads -- 3-4 us of a 36 us cheap instance when the wave is alone on its SIMD (the drain).  The bin counts
    // do not change while the kernel runs: every wave scans them once into LDS.  The next ticket is drawn at the top of phase E and resolved
    // behind its recursion, so both remaining trips run under the expansion (F20_TICKET_AHEAD=0: the plain f20_next between instances).
    int* const sch_incl = reinterpret_cast<int*>(lds_raw + FusedLds::oSch);
    int* const sch_cnt = sch_incl + 64;
#if F20_TICKET_AHEAD
    if (cap != 0) {
        LAUNDER_LANE(lane_s);
        const int c = sched[F20_BINS0 + F20_NB - 1 - lane_s];
        sch_cnt[lane_s] = c;
        sch_incl[lane_s] = wave_scan_incl_int(c);
        WSYNC();
    }
    // instance of ticket t (wave-uniform), -1 behind the batch: f20_next's walk over the bins, the counts from LDS
    auto ticket_entry_addr = [&](const int t, const int lane_) __attribute__((always_inline)) -> const int* {
        const int incl = sch_incl[lane_], c = sch_cnt[lane_];
        const unsigned long long m = __ballot(incl > t);
        if (m == 0ull) return nullptr;
        const int l = __ffsll((long long)m) - 1;
        const int base = __builtin_amdgcn_readlane(incl - c, l);
        return sched + F20_HDR + (size_t)(F20_NB - 1 - l) * cap + (t - base);
    };
    int next_inst = -2, next_tk = 0;                                    // -2: not drawn yet; next_tk: the ticket of next_inst
#endif
    // ---- expansion queue.  What follows the interior point -- phase A once more and phase E, ~15 of the ~54 us of an instance the trial
    // solves -- needs nothing of the instance but its step du (40 doubles): the linearisation is recomputed from the inputs anyway.  A wave
    // that has finished phase D of an instance therefore PUSHES the expansion (du into the per-instance buffer, the instance index into the
    // queue) and draws the next ticket at once; the expansions are popped by the waves that find the ticket queue dry.  The batch is then a
    // list of jobs whose last and most numerous ones take 15 us instead of 54: the ramp at the end of the launch (two jobs per wave slot at
    // B = 4096: a quarter of the run time with half of the slots idle) is filled with them.  Pops are fetch-adds (a compare-and-swap
    // on the pop counter serialised 2000 waves: 35 ms per step); what makes that safe is at the pop.  Visibility across the XCDs' L2s
    // without cache-wide write-backs: du and the queue entries are agent-scope relaxed atomic stores and loads (sc1: coherent per location
    // across the XCDs; as read-modify-write atomics the 80 accesses per expansion cost 70 us per step), ordered by the wave's own
    // s_waitcnt between them; the counters are fetch-adds like the ticket counter.
    // One queue would be one cache line of counters: same-line atomics retire at ~10 ns each, and the 15 us jobs of 2048 waves ask for more
    // pops than that (measured: the step 35 % SLOWER).  So F20_NQ sub-queues, a counter line each; ticket t pushes into queue (t - grid) mod
    // F20_NQ, which makes every queue's entry count known; a wave pops from the first queue at or behind its home (block mod F20_NQ) whose
    // pop counter -- one vector load over all queues -- is below its count.
    //   qh[q][32]: pushes at +0, pops at +1 (zeroed by the order kernel); qitems[q][qcap] = instance (-1 until published, -2: none); dubuf[inst][40]
    int* const qh = sched + F20_HDR + (size_t)F20_NB * cap;
    int* const qitems = qh + F20_NQ * 32;
    const int qcap = cap / F20_NQ + 1;
    double* const dubuf = reinterpret_cast<double*>(qitems + cap + F20_NQ + (cap & 1));
    const int nent = B - (int)gridDim.x;                                // entries of all queues: one per instance drawn by ticket
    const bool can_defer = F20_DEFER && cap != 0 && !park_gt;           // (a parked linearisation lives in the slot of the wave that shot it)
    (void)qcap; (void)dubuf; (void)nent; (void)can_defer;               // (unused with F20_DEFER == 0)
    for (;;) {
        LAUNDER_LANE(lane0);
#if F20_TICKET_AHEAD
        int inst;
        if (cap == 0) inst = first_ticket ? (int)blockIdx.x : -1;
        int cur_tk = next_tk;                                           // the ticket of this instance (its sub-queue)
        if (cap == 0) { }
        else if (next_inst != -2) inst = next_inst;
        else {
            int t = (int)blockIdx.x;
            if (!first_ticket) { int v = 0; if (lane0 == 0) v = atomicAdd(sched, 1); t = (int)gridDim.x + __builtin_amdgcn_readfirstlane(v); }
            const int* const e = ticket_entry_addr(t, lane0);
            inst = e ? __builtin_amdgcn_readfirstlane(*e) : -1;
            cur_tk = t;
        }
        next_inst = -2;
#else
        int inst = f20_next(sched, cap, first_ticket, lane0);
#endif
        bool ejob = false;
#if F20_DEFER
        if (inst < 0 && can_defer) {
            // the ticket queue is dry: pop an expansion.  Every instance drawn by ticket pushes exactly one entry (its expansion, or -2 when it
            // has none: failed, or skipped by its status), so every queue's number of entries is known.  A pop beyond it sends the wave to
            // the next queue (and out when all are spent); a pop in front of its push waits for it -- the pusher holds a ticket, so it is
            // a RUNNING wave (the workgroups that have not started yet hold first tickets only, and those never push): the wait cannot
            // depend on a workgroup that needs this wave's slot.
            const int home = (int)blockIdx.x & (F20_NQ - 1);
            for (;;) {
                const int ql = (lane0 + home) & (F20_NQ - 1);           // lane l looks at queue home + l
                const int tot = nent > ql ? (nent - ql + F20_NQ - 1) / F20_NQ : 0;
                const int popped = __hip_atomic_load(qh + ql * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long open = __ballot(popped < tot);
                if (open == 0ull) break;                                // every queue is spent: the wave leaves
                const int l = __ffsll((long long)open) - 1;
                const int q = (l + home) & (F20_NQ - 1);
                const int tq = __builtin_amdgcn_readlane(tot, l);
                int pq = 0;
                if (lane0 == 0) pq = atomicAdd(qh + q * 32 + 1, 1);
                pq = __builtin_amdgcn_readfirstlane(pq);
                if (pq >= tq) continue;                                 // somebody was faster: look again
                int i = -1;
                for (;;) {
                    if (lane0 == 0) i = __hip_atomic_load(qitems + q * qcap + pq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    i = __builtin_amdgcn_readfirstlane(i);
                    if (i != -1) break;
                    __builtin_amdgcn_s_sleep(16);
                }
                if (i < 0) continue;                                    // an instance without an expansion
                inst = i; ejob = true;
                next_inst = -1;                                         // the ticket queue stays dry
                break;
            }
        }
#endif
#if F20_CLASS_PRIO
        const bool first_class = first_ticket && (int)blockIdx.x < (int)gridDim.x / 2;
#endif
        const bool by_ticket = !first_ticket; (void)by_ticket; (void)cur_tk;
        first_ticket = false;
        if (inst < 0) break;
        F20_TRACE_BEGIN();
        auto push_none = [&]() __attribute__((always_inline)) {       // an instance drawn by ticket that has no expansion still owes the queue its entry
#if F20_DEFER
            if (can_defer && by_ticket && !ejob) {
                LAUNDER_LANE(lq);
                const int q = (cur_tk - (int)gridDim.x) & (F20_NQ - 1);
                if (lq == 0) { const int qs = atomicAdd(qh + q * 32, 1); __hip_atomic_store(qitems + q * qcap + qs, -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            }
#endif
        };
        if (!ejob && !first_pass && statusg[inst] != 0) { push_none(); continue; }    // failed / converged in an earlier SQP iteration of this call
        double* const xbg = xbarg + (size_t)inst * (N + 1) * NX;
        double* const ubg = ubarg + (size_t)inst * N * NU;
        const double* yrg = yrefg + (size_t)inst * N * NY;

        // =================================================================================================================
        // phase A (H0/H1): ERK4 + forward sensitivities.  Lane 3k + g <-> (stage k, column group g) as in kernel A:
        // g = 0: x-columns 2,3,4; g = 1: x-columns 5,6; g = 2: u-columns 0,1.  Lanes 60..63 shadow task 59 and store nothing.
        // =================================================================================================================
        double du = 0.0;
        bool failed = false;
        int it = 0;
        // without a slot buffer phase A runs a second time in front of phase E: pass 1 of this loop (one copy of the code)
        int npass = park_gt ? 1 : 2; asm volatile("" : "+s"(npass));
        bool deferred = false;
        int pass0 = 0;
#if F20_DEFER
        if (ejob) {                                                     // a popped expansion: its step, then pass 1 (phase A) and phase E
            LAUNDER_LANE(lq);
            du = lq < n ? __hip_atomic_load(dubuf + (size_t)inst * n + lq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            pass0 = 1;
        }
#endif
        asm volatile("" : "+s"(pass0));
#pragma unroll 1
        for (int pass = pass0; pass < npass; ++pass) {
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const double h = cf->Ts;
            const int tsk = lane < 3 * N ? lane : 3 * N - 1;
            const int k = (int)(((unsigned)tsk * 21846u) >> 16), g = tsk - 3 * k;      // tsk / 3, tsk % 3 without a narrow udivrem
            const bool live = lane < 3 * N;
        F20_STAMP(0);
            // ---- A1: the state through the four RK stages (all three lanes of a stage, redundantly: they share the GP sums);
            //      lane g == 0 tables the Jacobian entries of every RK stage and writes the defect
            {
                const double pin = pg[inst];
                double x[NX], u[NU], xn1[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) { x[i] = LDG(xbg + k * NX + i); xn1[i] = LDG(xbg + (k + 1) * NX + i); }
                u[0] = LDG(ubg + k * NU); u[1] = LDG(ubg + k * NU + 1);
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
                        double2* jt = reinterpret_cast<double2*>(JT + k * FusedLds::JTK + s * FusedLds::JTS);
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
        F20_STAMP(1);
            // ---- A2: the sensitivity columns of the lane's group from the tabled Jacobians (text of rk4_group / sens_rhs)
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
                    double S[3][NX];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                        for (int i = 0; i < NX; ++i) {
                            const double id = (g < 2 && i == xcol0 + cc) ? 1.0 : 0.0;
                            S[cc][i] = id + cs * h * kS[cc][i];
                        }
                    ModelEvalT<double> e;
                    {
                        const double2* jt = reinterpret_cast<const double2*>(JT + k * FusedLds::JTK + s * FusedLds::JTS);
                        double2 q;
                        q = jt[0]; e.j0[0] = q.x; e.j0[1] = q.y; q = jt[1]; e.j0[2] = q.x; e.j1[0] = q.y; q = jt[2]; e.j1[1] = q.x; e.j1[2] = q.y;
#pragma unroll
                        for (int r = 0; r < 3; ++r) { q = jt[3 + 2 * r]; e.a[r][0] = q.x; e.a[r][1] = q.y; q = jt[4 + 2 * r]; e.a[r][2] = q.x; e.a[r][3] = q.y; }
#pragma unroll
                        for (int r = 0; r < 3; ++r) { q = jt[9 + r]; e.bu[r][0] = q.x; e.bu[r][1] = q.y; }
                    }
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        sens_rhs<double>(e, S[cc], g == 2 ? cc : -1, kS[cc]);
#pragma unroll
                        for (int i = 0; i < NX; ++i) accS[cc][i] += ws * kS[cc][i];
                    }
                }
                // packed stage record: stored columns c = 0..6 <-> (A[:,2..6], B[:,0..1]), rows 0..5 of each
                const int c0 = g == 0 ? 0 : (g == 1 ? 3 : 5);
                const int nc = g == 0 ? 3 : 2;
                WSYNC();                                   // REQUIRED: GT overlays the Jacobian tables (FusedLds: oGTC = oJT = 0) -- every lane has read its
                                                           // tables before the first store.  With ADMPC_WSYNC_FENCE_ONLY this is a wave-level fence: the
                                                           // kernel must stay ONE wave per workgroup (launch bounds below)
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
            if (park_gt) {      // park the linearisation in the wave's slot buffer (read back in front of phase E); the stores retire under phase C
                stage_in<N * GTS>(slot, GT, lane);
                stage_in<N * NX>(slot + N * GTS, bl, lane);
            }
        }
        if (pass != 0) break;

        // =================================================================================================================
        // phase C (H2-H4): condensing, lane i <-> input i = 2k + j (text of admpc_condense_kernel).  Leaves the packed
        // Hessian rows in Hp and, in registers, g0 (reduced gradient at du = 0) and xhat6 of the lane's stage.
        // =================================================================================================================
        F20_STAMP(2);
        double g0, xh6_own = 0.0;
#if F20_MFMA == 2
        // ---- condensing on the matrix pipe.  Gamma_k (7 x 40) and the free response xhat_k (as a 41st column) live in three
        // 16-column MFMA tiles in the C/D register layout (lane & 15 = column inside the block, (lane >> 4) + 4 v = row): register v of a
        // tile IS the B operand of K-step v of the next product, so the stage recursion Gamma_{k+1} = A_k Gamma_k runs tile -> tile with
        // no exchange at all, and the same registers are both operands of H += (W Gamma)' Gamma.  Column 40 carries xhat_k through the
        // recursion (+ b_k) and, with the tracking error added, makes row 40 of the accumulated product the reduced gradient:
        // g0_i = r_i + sum_k Gamma_k[:, i]' W (xhat_k + xbar_k - xref_k).  Per stage the vector pipe only fetches the operands (A_k, B_k,
        // b_k, the error from LDS) and selects; 184 v_mfma_f64_16x16x4_f64 per instance replace ~3400 vector instructions.
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const int ji = lane & 1;
            const bool uact = lane < n;
            const int sc = uact ? lane : 0;
            const int r16 = lane & 15, kq = lane >> 4;
            const double Ts = cf->Ts, h = cf->Ts;
            const double Rj = Ts * cf->W[NX + ji];
            stage_dq_nt<N>(dqC, xbg, yrg, yrefeg + (size_t)inst * NX, lane);
            const double ubar_i = ubg[sc];
            const double r_i = Rj * (ubar_i - yrg[(sc >> 1) * 9 + 7 + (sc & 1)]);
            // weights of the rows this lane holds: row kq (v = 0) and row 4 + kq (v = 1; row 7 does not exist)
            const double wq0 = Ts * cf->W[kq], we0 = cf->We[kq];
            const double wq1 = kq < 3 ? Ts * cf->W[4 + (kq < 3 ? kq : 0)] : 0.0, we1 = kq < 3 ? cf->We[4 + (kq < 3 ? kq : 0)] : 0.0;
            const bool any_hi = (QMASK >> 4) != 0;                      // a tracking weight on v_y, psi_dot or delta: second K-step of H
            const bool is40 = r16 == 8;                                 // (tile 2 only) the free-response column
            const d4 z4 = {0.0, 0.0, 0.0, 0.0};
            d4 G[3] = { z4, z4, z4 };
            G[2][0] = is40 ? x0g[(size_t)inst * NX + kq] - xbg[kq] : 0.0;                                          // xhat_0 = x0 - xbar_0
            G[2][1] = (is40 && kq < 3) ? x0g[(size_t)inst * NX + 4 + (kq < 3 ? kq : 0)] - xbg[4 + (kq < 3 ? kq : 0)] : 0.0;
            d4 acc[3][3];
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J < 3; ++J) acc[I][J] = z4;
            WSYNC();
            static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int lim = 2 * k < n ? 2 * k : n;        // inputs of stages < k: the non-zero columns of Gamma_k
                constexpr int nblk = (lim + 15) / 16;             // column blocks that hold them
                int tok = B; asm volatile("" : "+s"(tok));         // one stage = one basic block (see the vector version)
                if (tok > 0) {
                if constexpr (k >= 1) {
                    // ---- cost of stage k: H += (W Gamma_k)' Gamma_k, with the tracking error on the free-response column
                    const double e0 = dqC[k * 7 + kq], e1 = dqC[k * 7 + 4 + (kq < 3 ? kq : 0)];
                    d4 Gh2 = G[2];
                    Gh2[0] += is40 ? e0 : 0.0;
                    Gh2[1] += (is40 && kq < 3) ? e1 : 0.0;
                    if constexpr (k < N) { const double x6 = rdlane(G[2][1], 40); if (lane == k) xh6_own = x6; }      // xhat_k[6]: row 6 = 2 + 4 * 1 of column 40
                    const double wl0 = k < N ? wq0 : we0, wl1 = k < N ? wq1 : we1;
                    static_for<0, nblk>([&](auto Jc) __attribute__((always_inline)) {
                        constexpr int J = decltype(Jc)::value;
                        static_for<J, 3>([&](auto Ic) __attribute__((always_inline)) {
                            constexpr int I = decltype(Ic)::value;
                            if constexpr (I < nblk || I == 2) {
                                const d4& GI = I == 2 ? Gh2 : G[I];
                                const d4& GJ = J == 2 ? Gh2 : G[J];
                                acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(wl0 * GI[0], GJ[0], acc[I][J], 0, 0, 0);
                                if (any_hi) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(wl1 * GI[1], GJ[1], acc[I][J], 0, 0, 0);
                            }
                        });
                    });
                }
                if constexpr (k < N) {
                    // ---- propagate: Gamma_{k+1} = A_k Gamma_k, xhat_{k+1} = A_k xhat_k + b_k.  A operand of K-step s: A_k[lane & 15][4 s + (lane >> 4)]
                    // (columns 0, 1 of A are unit vectors, row 6 is e6: not stored in the packed record)
                    const double* Gk = GT + k * GTS;
                    const int rr = r16 < 6 ? r16 : 0;
                    const double a0l = Gk[(kq >= 2 ? kq - 2 : 0) * 6 + rr], a1l = Gk[((kq < 3 ? kq : 0) + 2) * 6 + rr];
                    const double Aop0 = kq < 2 ? (r16 == kq ? 1.0 : 0.0) : (r16 < 6 ? a0l : 0.0);
                    const double Aop1 = kq < 3 ? (r16 < 6 ? a1l : ((r16 == 6 && kq == 2) ? 1.0 : 0.0)) : 0.0;
                    d4 Gn[3] = { z4, z4, z4 };
                    static_for<0, 3>([&](auto Jc) __attribute__((always_inline)) {
                        constexpr int J = decltype(Jc)::value;
                        if constexpr (J < nblk || J == 2)
                            Gn[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop0, G[J][0], __builtin_amdgcn_mfma_f64_16x16x4f64(Aop1, G[J][1], z4, 0, 0, 0), 0, 0, 0);
                    });
                    Gn[2][0] += is40 ? bl[k * 7 + kq] : 0.0;
                    Gn[2][1] += (is40 && kq < 3) ? bl[k * 7 + 4 + (kq < 3 ? kq : 0)] : 0.0;
                    // the two inputs of stage k enter with B_k: columns 2k, 2k + 1 of block (2k) / 16, rows 0..5 from the record, row 6 = (0, h)
                    constexpr int Jn = (2 * k) / 16;
                    const bool bsel = (r16 >> 1) == (k & 7);
                    const int jc = r16 & 1;
                    const double b0l = Gk[(5 + jc) * 6 + kq], b1l = Gk[(5 + jc) * 6 + 4 + (kq < 2 ? kq : 0)];
                    const double b1v = kq < 2 ? b1l : (kq == 2 ? (jc ? h : 0.0) : 0.0);
                    Gn[Jn][0] = bsel ? b0l : Gn[Jn][0];
                    Gn[Jn][1] = bsel ? b1v : Gn[Jn][1];
#pragma unroll
                    for (int J = 0; J < 3; ++J) G[J] = Gn[J];
                }
                }
            });
            // the tiles into the packed lower-triangular rows: element (row = lane >> 4 + 4 v, column = lane & 15) of tile (I, J);
            // row 40 (tile row block 2, element 8 = 0 + 4 * 2: lanes 0..15, v = 2) is the reduced gradient without its input part
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * I + kq + 4 * v, col = 16 * J + r16;
                        if (row < n && col <= row) Hp[row * (row + 1) / 2 + col] = acc[I][J][v];
                    }
            if (kq == 0) { gam[r16] = acc[2][0][2]; gam[16 + r16] = acc[2][1][2]; gam[32 + r16] = acc[2][2][2]; }
            WSYNC();
            g0 = r_i + gam[sc];
            if (uact) Lp[(lane * (lane + 1)) / 2 + lane] = 0.0;         // diagonal slots of the packed factor (see the vector version)
            WSYNC();
        }
#else
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const int ki = lane >> 1, ji = lane & 1;
            const bool uact = lane < n;
            const int sc = uact ? lane : 0;
            const double Ts = cf->Ts, h = cf->Ts;
            const double Rj = Ts * cf->W[NX + ji];
            stage_dq_nt<N>(dqC, xbg, yrg, yrefeg + (size_t)inst * NX, lane);
            const double ubar_i = ubg[sc];
            const double r_i = Rj * (ubar_i - yrg[(sc >> 1) * 9 + 7 + (sc & 1)]);
            double Qd[NX], Qe[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { Qd[i] = Ts * cf->W[i]; Qe[i] = cf->We[i]; }
            // the carried columns in two sets used alternately (stage k reads set k & 1 and writes the other one): with one set the
            // results of a stage were copied into it behind every stage (the stages are separate basic blocks: ten v_mov_b64 each)
            double xh2[2][NX], g2[2][NX];
#pragma unroll
            for (int c = 0; c < NX; ++c) { xh2[0][c] = x0g[(size_t)inst * NX + c] - xbg[c]; xh2[1][c] = 0.0; }      // uniform
            WSYNC();
#pragma unroll
            for (int c = 0; c < NX; ++c) { g2[0][c] = (F20_XHLANE && lane == n) ? xh2[0][c] : 0.0; g2[1][c] = 0.0; }      // lane 40: xhat_0 = x0 - xbar_0
#if F20_MFMA
            // H on the matrix pipe: six 16 x 16 tiles (I >= J) of v_mfma_f64_16x16x4_f64.  One MFMA step takes K = 4 rows of the
            // 60 x 40 matrix G whose rows are the weighted components of Gamma_k: per stage the components of QMASK, four at a time
            // (QMASK 7: x, y, psi + a zero row).  Operand layout: lane = index + 16 * k  ->  lane >> 4 picks the component, lane & 15
            // the column inside the block; operands come from the LDS exchange buffer the vector version broadcasts from.  The matrix
            // pipe is otherwise idle and runs under the propagation FMAs: 1260 v_fmac_f64_dpp per instance leave the vector pipe
            // (measured in isolation: 13.4 -> 10.1 us per instance, profiles/r3/mfma_condense_ab.txt).
            constexpr int NCOMP = ((QMASK >> 0) & 1) + ((QMASK >> 1) & 1) + ((QMASK >> 2) & 1) + ((QMASK >> 3) & 1) + ((QMASK >> 4) & 1) + ((QMASK >> 5) & 1) + ((QMASK >> 6) & 1);
            constexpr int NSTEP = (NCOMP + 3) / 4;
            const int r16 = lane & 15, kq = lane >> 4;
            int crow[NSTEP]; double wq_l[NSTEP], we_l[NSTEP];       // per MFMA step: this lane's component (7 = the zero row) and its weights
            {
                int seen = 0;
#pragma unroll
                for (int st_ = 0; st_ < NSTEP; ++st_) { crow[st_] = 7; wq_l[st_] = 0.0; we_l[st_] = 0.0; }
                static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr ((QMASK >> c) & 1) {
                        const int st_ = seen >> 2, kk = seen & 3;          // compile-time after unrolling
#pragma unroll
                        for (int q_ = 0; q_ < NSTEP; ++q_) if (q_ == st_ && kq == kk) { crow[q_] = c; wq_l[q_] = Qd[c]; we_l[q_] = Qe[c]; }
                        ++seen;
                    }
                });
            }
            gam[7 * 64 + lane] = 0.0;                                // the zero row (gam has room for an eighth row)
            d4 acc[3][3];
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J < 3; ++J) acc[I][J] = d4{0.0, 0.0, 0.0, 0.0};
#else
            double hrow[n];
#pragma unroll
            for (int i = 0; i < n; ++i) hrow[i] = 0.0;
#endif
            g0 = r_i;
            static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int lim = 2 * k < n ? 2 * k : n;        // inputs of stages < k (even)
                constexpr int nblk = (lim + 15) / 16;             // 16-lane blocks of Gamma that are non-zero at this stage
                double (&g)[NX] = g2[k & 1]; double (&xh)[NX] = xh2[k & 1];
                double (&gn)[NX] = g2[(k + 1) & 1]; double (&xn)[NX] = xh2[(k + 1) & 1];
                // One stage = one basic block (an always-true test the compiler cannot see through): merged into one 21-stage block,
                // hipcc hoists every LDS load of the whole instance and spills ~1600 registers.
                int tok = B; asm volatile("" : "+s"(tok));
#if F20_TOKTRAP
                // the never-taken side of the test ends the program instead of joining the stage: no phi nodes behind the stage, i.e. no
                // copies of the carried columns (g, xh: ten v_mov_b64 per stage) into the registers the skipped path would have kept
                if (tok <= 0) __builtin_trap();
                {
#else
                if (tok > 0) {
#endif
                double wg[NX];
#if F20_MFMA
                double blk[NSTEP][3];
#else
                double Rb[NX][3];
#endif
#if F20_CPREF
                // Every LDS read of the stage's record is issued before anything else of the stage: left to itself hipcc keeps one or two
                // ds_read_b128 in flight and the propagation (60 multiply-adds behind 21 reads) waits on each of them in turn.
                double2 Av[15], Bv[3]; double blv[NX], Aq[2]; (void)Av; (void)Aq;
                const bool mine_p = ki == k;
                if constexpr (k < N) {
                    const double* Gk = GT + k * GTS;
                    // (indices into the one LDS array, not selected pointers: hipcc turns a select of two LDS pointers into flat pointers and
                    // converts every element address back with a null check -- five scalar instructions per load)
                    const int bidx = mine_p ? FusedLds::oGTC + k * GTS + 5 * 6 + 6 * ji : FusedLds::oGam + 7 * 64;
#if F20_ADPP
                    { const int e16 = lane & 15; Aq[0] = Gk[e16]; Aq[1] = Gk[16 + (e16 < 14 ? e16 : 13)]; }
#else
#pragma unroll
                    for (int q_ = 0; q_ < 15; ++q_) Av[q_] = *reinterpret_cast<const double2*>(Gk + 2 * q_);
#endif
#pragma unroll
                    for (int q_ = 0; q_ < 3; ++q_) Bv[q_] = *reinterpret_cast<const double2*>(lds_raw + bidx + 2 * q_);
#if F20_XHLANE
                    const int lidx = lane == n ? FusedLds::oBlA + k * 7 : FusedLds::oGam + 7 * 64;      // b_k for the column of the free response, zeros for the others
#pragma unroll
                    for (int r = 0; r < NX; ++r) blv[r] = lds_raw[lidx + r];
#else
#pragma unroll
                    for (int r = 0; r < NX; ++r) blv[r] = bl[k * 7 + r];
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
                if constexpr (k >= 1) {
#if F20_XHLANE
                    static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr ((QMASK >> c) & 1) {
                            const double w = k < N ? Qd[c] : Qe[c];
                            wg[c] = w * g[c];
                            gam[c * 64 + lane] = g[c];
                        }
                    });
                    if constexpr (!((QMASK >> 6) & 1)) gam[6 * 64 + lane] = g[6];      // xhat_k[6] for the steering rows
                    static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr ((QMASK >> c) & 1) g0 += wg[c] * (gam[c * 64 + n] + dqC[k * 7 + c]);      // lane 40 has just published xhat_k[c]
                    });
                    { double x6 = gam[6 * 64 + n]; asm volatile("" : "+v"(x6)); xh6_own = lane == k ? x6 : xh6_own; }      // (an unconditional load)
#else
                    if (lane == k) xh6_own = xh[6];
                    static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr ((QMASK >> c) & 1) {
                            const double w = k < N ? Qd[c] : Qe[c];
                            wg[c] = w * g[c];
                            g0 += wg[c] * (xh[c] + dqC[k * 7 + c]);
                            gam[c * 64 + lane] = g[c];
                        }
                    });
#endif
#if F20_MFMA
#pragma unroll
                    for (int st_ = 0; st_ < NSTEP; ++st_)
#pragma unroll
                        for (int m = 0; m < nblk; ++m) blk[st_][m] = gam[crow[st_] * 64 + 16 * m + r16];
#else
                    static_for<0, NX>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr ((QMASK >> c) & 1) {
#pragma unroll
                            for (int m = 0; m < nblk; ++m) Rb[c][m] = gam[c * 64 + 16 * m + (lane & 15)];
                        }
                    });
#endif
                }
                if constexpr (k < N) {
                    const double* Gk = GT + k * GTS; (void)Gk;
                    // The inputs of stage k enter with B_k: lanes 2k, 2k + 1 (whose column of Gamma is still zero) start the product from
                    // their column of B_k, every other lane from the zero row of gam -- one per-lane LDS address instead of 28 selects
                    // per stage (560 vector instructions per instance); 0 + A_k 0 = 0 exactly: the same bits as the selects gave.
                    const bool mine = ki == k;
#if F20_CPREF
#pragma unroll
                    for (int r = 0; r < 6; r += 2) { gn[r] = Bv[r / 2].x; gn[r + 1] = Bv[r / 2].y; }
                    gn[0] += g[0]; gn[1] += g[1];
                    gn[6] = mine ? (ji ? h : 0.0) : g[6];
#if F20_XHLANE
#pragma unroll
                    for (int r = 0; r < NX; ++r) gn[r] += blv[r];                  // b_k in lane 40, + 0.0 elsewhere
                    (void)xn; (void)xh;
#if F20_ADPP
                    static_for<0, 5>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        static_for<0, 6>([&](auto rc) __attribute__((always_inline)) {
                            constexpr int r = decltype(rc)::value, q_ = c * 6 + r;
                            fmac_rowbc_ld<q_ % 16>(gn[r], Aq[q_ / 16], g[c + 2]);      // gn[r] += A_k[r][c + 2] g[c + 2]
                        });
                    });
#else
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
#pragma unroll
                        for (int r = 0; r < 6; r += 2) {
                            const double2 a = Av[c * 3 + r / 2];
                            gn[r] += a.x * g[c + 2];  gn[r + 1] += a.y * g[c + 2];
                        }
                    }
#endif
#else
#pragma unroll
                    for (int r = 0; r < 6; ++r) xn[r] = r < 2 ? blv[r] + xh[r] : blv[r];
                    xn[6] = blv[6] + xh[6];
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
#pragma unroll
                        for (int r = 0; r < 6; r += 2) {
                            const double2 a = Av[c * 3 + r / 2];
                            xn[r] += a.x * xh[c + 2]; xn[r + 1] += a.y * xh[c + 2];
                            gn[r] += a.x * g[c + 2];  gn[r + 1] += a.y * g[c + 2];
                        }
                    }
#endif
#else
                    const double* const bsrc = mine ? Gk + 5 * 6 + 6 * ji : gam + 7 * 64;
#pragma unroll
                    for (int r = 0; r < 6; r += 2) {
                        const double2 v = *reinterpret_cast<const double2*>(bsrc + r);
                        gn[r] = v.x; gn[r + 1] = v.y;
                    }
                    gn[0] += g[0]; gn[1] += g[1];
                    gn[6] = mine ? (ji ? h : 0.0) : g[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) xn[r] = r < 2 ? bl[k * 7 + r] + xh[r] : bl[k * 7 + r];
                    xn[6] = bl[k * 7 + 6] + xh[6];
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
#pragma unroll
                        for (int r = 0; r < 6; r += 2) {
                            const double2 a = *reinterpret_cast<const double2*>(Gk + c * 6 + r);
                            xn[r] += a.x * xh[c + 2]; xn[r + 1] += a.y * xh[c + 2];
                            gn[r] += a.x * g[c + 2];  gn[r + 1] += a.y * g[c + 2];
                        }
                    }
#endif
                }
#if F20_MFMA
                if constexpr (k >= 1) {
                    // H[I][J] += (W G_I)' G_J for the tiles whose columns are non-zero already (inputs of stages < k): I, J < nblk, I >= J
#pragma unroll
                    for (int st_ = 0; st_ < NSTEP; ++st_) {
                        const double wl = k < N ? wq_l[st_] : we_l[st_];
                        static_for<0, nblk>([&](auto Ic) __attribute__((always_inline)) {
                            constexpr int I = decltype(Ic)::value;
                            const double a = wl * blk[st_][I];
                            static_for<0, I + 1>([&](auto Jc) __attribute__((always_inline)) {
                                constexpr int J = decltype(Jc)::value;
                                acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, blk[st_][J], acc[I][J], 0, 0, 0);
                            });
                        });
                    }
                }
#else
                if constexpr (k >= 1) {
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
#endif
                }
            });
#if F20_MFMA
            // the tiles into the packed lower-triangular rows: element (row = lane >> 4 + 4 v, column = lane & 15) of tile (I, J)
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * I + kq + 4 * v, col = 16 * J + r16;
                        if (row < n && col <= row) Hp[row * (row + 1) / 2 + col] = acc[I][J][v];
                    }
#else
            store_row_40(hrow, lds_byte_addr(Hp + (uact ? (lane * (lane + 1)) / 2 : 0)));
#endif
            // diagonal slots of the packed factor: 0.0 (the factorisation stores the strictly-lower part only; the substitution
            // assembly lets the source lane of a step take part with this multiplier).  Phase C used L's space: rewrite them.
            WSYNC();
            if (uact) Lp[(lane * (lane + 1)) / 2 + lane] = 0.0;
            WSYNC();
        }
#endif

        // =================================================================================================================
        // phase D (H5): unconstrained trial + interior point on the condensed QP (text of admpc_qp_dense_kernel)
        // =================================================================================================================
        F20_STAMP(3);
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const int ki = lane >> 1, ji = lane & 1;
            const bool uact = lane < n;
            const bool dact = lane >= 1 && lane < N;
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
            const double inv_nineq = 1.0 / (double)(8 * N + 2 * (N - 1));
            const double ubar_i = ubg[sc];
            const double dl_i = cf->lbu[ji] - ubar_i, duu_i = cf->ubu[ji] - ubar_i;
            double t[4], lam[4], sl = thr, su = thr;
            {
                const double r0[4] = { thr - dl_i, thr + duu_i, thr, thr };
#pragma unroll
                for (int i = 0; i < 4; ++i) { t[i] = r0[i] > thr ? r0[i] : thr; lam[i] = mu0 * rcp_nr(t[i]); }
            }
            double Dt[2] = {1.0, 1.0}, Dlam[2] = {0.0, 0.0}, Ddl = 0.0, Ddu = 0.0, dx6 = 0.0;
            if (dact) {
                const double x6 = xbg[lane * 7 + 6];
                Ddl = cf->lbx_delta - x6; Ddu = cf->ubx_delta - x6;
                dx6 = xh6_own;
                const double r0[2] = { dx6 - Ddl, Ddu - dx6 };
#pragma unroll
                for (int i = 0; i < 2; ++i) { Dt[i] = r0[i] > thr ? r0[i] : thr; Dlam[i] = mu0 * rcp_nr(Dt[i]); }
            }
            PK_DL = dl_i; PK_DUU = duu_i; PK_G0 = g0; PK_DDL = Ddl; PK_DDU = Ddu;      // parked in LDS: registers are the scarce resource
            WSYNC();

            double rmax_prev = 0.0, step = 1e300, stp_local = 1e300, alpha_prev = 1.0;
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
        F20_STAMP(4);
            F20_TRACE_MID();
            if (!solved)
            for (; it < itmax + (cons ? fbit : 0); ++it) {
                // An instance that is still iterating is on its way to becoming the batch's straggler: give its wave the issue slots of the
                // SIMD it shares (the partner is a fresh instance of the second round, which is not on anybody's critical path)
#ifndef F20_NO_PRIO
#if F20_CLASS_PRIO == 1        // experiment: the first-round first wave of a SIMD (predicted hardest 1024) one level above its partner
                if (first_class) { if (it == 0) __builtin_amdgcn_s_setprio(2); if (it == 2) __builtin_amdgcn_s_setprio(3); }
                else { if (it == 0) __builtin_amdgcn_s_setprio(1); if (it == 3) __builtin_amdgcn_s_setprio(2); if (it == 6) __builtin_amdgcn_s_setprio(3); }
#elif F20_CLASS_PRIO == 2      // experiment: first class at 3 from its first iteration, the others capped at 2
                if (first_class) { if (it == 0) __builtin_amdgcn_s_setprio(3); }
                else { if (it == 0) __builtin_amdgcn_s_setprio(1); if (it == 3) __builtin_amdgcn_s_setprio(2); }
#else
                if (it == F20_PRIO_IT0) __builtin_amdgcn_s_setprio(1);
                if (it == F20_PRIO_IT1) __builtin_amdgcn_s_setprio(2);
                if (it == F20_PRIO_IT2) __builtin_amdgcn_s_setprio(3);
#endif
#endif
                int lz = lane;                          // laundered lane id: per-lane addresses / predicates derived from it are recomputed in
                asm volatile("" : "+v"(lz));            // place instead of being hoisted out of the loops
                const int trz = lz * (lz + 1) / 2;
                const bool uz = lz < n;
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
                    cb[lane] = uact ? du : 0.0;
                    const double dlam_pref = wave_scan_incl<OpSum>(dact ? (Dlam[1] - Dlam[0]) : 0.0);     // lanes = stages
                    sb[lane] = rdlane(dlam_pref, 63) - dlam_pref;           // suffix over stages > lane
                    const double Ssuf_incl = wave_scan_incl<OpSum>(dact ? G56 : 0.0);
                    sb2[lane] = rdlane(Ssuf_incl, 63) - Ssuf_incl;          // lane = stage: sum over stages > lane
                    WSYNC();
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
                    cmax = wave_reduce<OpMax0>(cmax);
                    rmax = wave_reduce<OpMaxNan0>(rmax);
                    step = wave_reduce<OpMax0>(stp_local);
#ifdef F20_DEBUG
                    if (lane == 0) printf("[f20] it %2d mu %.3e cmax %.3e rmax %.3e rmax_prev %.3e step %.3e alpha_prev %.6f\n", it, mu, cmax, rmax, rmax_prev, step, alpha_prev);
#endif
                    if (!(mu == mu) || !(rmax == rmax)) { failed = true; break; }
                    if (cmax <= tol_comp && step <= tol_step &&
                        (rmax <= tol_res || (it > 0 && rmax > 0.1 * rmax_prev && rmax <= ADMPC_IPM_FLOOR_CAP * tol_res))) break;      // admpc.h: stopping test
                    rmax_prev = rmax;
                }
                if (fbit > 0 && !cons && it >= fbit) {
                    cons = true; warmed = false;
                    cold_start();
                    rmax_prev = 0.0;
                    --it;
                    continue;
                }
                factorise(Dbar, (uz && ji) ? S_i : 0.0, lz);
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
                    const double x = ldl_solve(y, lz);
                    ddu = uact ? x : 0.0;
                    cb[lane] = ddu;
                    WSYNC();
                    const double du1_stage = lane < N ? cb[2 * lane + 1] : 0.0;
                    const double pre = wave_scan_incl<OpSum>(du1_stage);
                    const double ddx6 = h * (pre - du1_stage);
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
                    rr = wave_reduce<OpMax0>(rr);
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
                            warmed = false;
                            cold_start();
                        } else {
                        alpha_prev = alpha;
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
                        }
                    }
                    WSYNC();
                }
            }
        }
        F20_STAMP(5);
        if (failed) break;
#if F20_DEFER
        if (can_defer && by_ticket && !ejob) {
            // push the expansion; the next ticket and the queue slot travel together with the stores of du
            LAUNDER_LANE(lq);
            int tk = 0, qs = 0;
            if (lq == 0) tk = atomicAdd(sched, 1);
            if (lq < n) __hip_atomic_store(dubuf + (size_t)inst * n + lq, du, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int q = (cur_tk - (int)gridDim.x) & (F20_NQ - 1);
            if (lq == 0) qs = atomicAdd(qh + q * 32, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // du is at the memory side before the entry can be seen
            if (lq == 0) __hip_atomic_store(qitems + q * qcap + qs, inst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            next_tk = (int)gridDim.x + __builtin_amdgcn_readfirstlane(tk);
            const int* const e = ticket_entry_addr(next_tk, lq);
            next_inst = e ? __builtin_amdgcn_readfirstlane(*e) : -1;
            deferred = true;
            break;
        }
#endif
        }   // pass
        if (!ejob) { LAUNDER_LANE(lw); if (lw == 0 && itersg) itersg[inst] = it; }
        if (deferred) { __builtin_amdgcn_s_setprio(0); continue; }
        if (failed) {
            // non-finite QP data: acados returns before the update -- the iterate stays as it is, status 4, cost +inf
            { LAUNDER_LANE(lw); if (lw == 0) { statusg[inst] = ADMPC_STATUS_QP_FAILURE; if (costg) costg[inst] = INFINITY; } }
            push_none();
            WSYNC();
            __builtin_amdgcn_s_setprio(0);
            continue;
        }

        // =================================================================================================================
        // phase E (H6): expand the states through the linearised dynamics, full step, cost, status (text of admpc_expand_kernel)
        // =================================================================================================================
        {
            LAUNDER_LANE(lane); LAUNDER_CFG(cf);
            const int ji = lane & 1;
            const bool uact = lane < n;
            const int sc = uact ? lane : 0;
            const double Ts = cf->Ts, h = cf->Ts;
            const double Rj = Ts * cf->W[NX + ji];
            const double rho_l = Ts * cf->zl, rho_u = Ts * cf->zu;
            const int r6 = lane < 6 ? lane : 0;                     // row of the packed linearisation this lane reads
            const int r7 = lane < NX ? lane : 0;
            const double wq = lane < NX ? Ts * cf->W[r7] : 0.0, wqe = lane < NX ? cf->We[r7] : 0.0;
#if F20_TICKET_AHEAD
            int tk_v = 0;
            if (cap != 0 && !ejob && lane == 0) tk_v = atomicAdd(sched, 1);       // the next ticket: its trip to the L2 runs under the expansion
#endif
            WSYNC();
            if (park_gt) {
                // the slot buffer again: the wave's own stores of phase A have long retired, but the CU's vector L1 may still hold the
                // lines the PREVIOUS instance of this wave read here (the L1 does not follow the wave's stores): drop them
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                stage_in<N * GTS>(GTe, slot, lane);
                stage_in<N * NX>(ble, slot + N * GTS, lane);
                WSYNC();
            }
            stage_dq_nt<N>(dqE, xbg, yrg, yrefeg + (size_t)inst * NX, lane);
            du = uact ? du : 0.0;
            dus[lane] = du;
            const double ubar_i = ubg[sc];
            const double uref_i = yrg[(sc >> 1) * 9 + 7 + (sc & 1)];
            double dx = lane < NX ? x0g[(size_t)inst * NX + r7] - xbg[r7] : 0.0;      // dx_0 (lanes 0..6)
            WSYNC();
            bool bad = false;
            double J = 0.0;
            static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const double e = dx + dqE[k * 7 + r7];
                J += 0.5 * (k < N ? wq : wqe) * e * e;
                if (!(fabs(dx) <= 1e300)) bad = true;
                if (lane < NX) dqE[k * 7 + lane] = dx;            // slot k now holds dx_k
                if constexpr (k < N) {
                    const double* Gk = GTe + k * GTS;
                    const double u0 = dus[2 * k], u1 = dus[2 * k + 1];
                    // rows 0..5 from the packed record; every lane loads (lanes >= 6 read row 0 and drop the result): seven loads under their
                    // own EXEC masks were 18 scalar instructions per stage.  Row 6 of [A B] is [e6, 0, h]: one multiply-add in lane 6.
                    const double bk = ble[k * 7 + r7];
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
#if F20_TICKET_AHEAD
            int tk_e = -1;                                                // the list entry of the next ticket: loaded under the output stores
            if (cap != 0 && !ejob) {
                next_tk = (int)gridDim.x + __builtin_amdgcn_readfirstlane(tk_v);
                const int* const e = ticket_entry_addr(next_tk, lane);
                if (e) tk_e = *e;
            }
#endif
            const double unew = ubar_i + du;
            if (uact && !(fabs(unew) <= 1e300)) bad = true;
            const int status = __any(bad) ? ADMPC_STATUS_QP_FAILURE : ADMPC_STATUS_SUCCESS;
            double Ju = 0.0;
            WSYNC();
            if (status == 0) {
#pragma unroll
                for (int i0 = 0; i0 < (N + 1) * NX; i0 += WAVE) { const int i = i0 + lane; if (i < (N + 1) * NX) STG(xbg + i, LDG(xbg + i) + dqE[i]); }
                if (uact) {
                    const double e = unew - uref_i;
                    Ju = 0.5 * Rj * e * e;
                    if (unew < cf->lbu[ji]) Ju += rho_l * (cf->lbu[ji] - unew);
                    if (unew > cf->ubu[ji]) Ju += rho_u * (unew - cf->ubu[ji]);
                    STG(ubg + lane, unew);
                }
            }
            const double Jt = wave_reduce<OpSum>(J + Ju);
            if (lane == 0) {
                if (costg) costg[inst] = status == 0 ? Jt : INFINITY;
                statusg[inst] = status;
            }
            WSYNC();
#if F20_TICKET_AHEAD
            if (cap != 0 && !ejob) next_inst = __builtin_amdgcn_readfirstlane(tk_e);
#endif
            F20_STAMP(6);
            F20_TRACE_END(inst);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    F20_FLUSH();
    // ---- every wave has drawn exactly one ticket beyond the batch; the last one to leave clears tickets and bins for the next launch
    {
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
}

}  // namespace

// ---- host side (called by admpc_solve_batch_ex in admpc_kernels.hip)
extern "C" {

__attribute__((visibility("hidden"))) int admpc_fused20_lds_bytes(void) { return FusedLds::total * (int)sizeof(double); }
// doubles of the per-wave slot buffers (only allocated and passed when the handle's model carries GP residuals)
__attribute__((visibility("hidden"))) size_t admpc_fused20_slot_doubles(int num_cu) { return (size_t)num_cu * 8 * FusedLds::SLOT; }

// debug builds only: read and clear the phase counters (all zero in the shipped build)
#ifdef F20_ORDER_HINT
int admpc_debug_f20_order_hint(const int* d_hint) { return hipMemcpyToSymbol(HIP_SYMBOL(g_order_hint), &d_hint, sizeof(d_hint)) == hipSuccess ? 0 : -1; }
#endif
int admpc_debug_f20_ticks(unsigned long long* out16)
{
#ifdef ADMPC_PHASE_TIMERS
    unsigned long long z[16] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_f20_ticks), sizeof z) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_f20_ticks), z, sizeof z) != hipSuccess) return -1;
    return 0;
#else
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    return 1;
#endif
}

int admpc_debug_f20_trace(unsigned long long* out, int n_inst)
{
#if defined(ADMPC_PHASE_TIMERS) || defined(ADMPC_F20_TRACE)
    if (n_inst > 8192) n_inst = 8192;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f20_trace), (size_t)n_inst * 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return 0;
#else
    (void)out; (void)n_inst;
    return 1;
#endif
}

__attribute__((visibility("hidden"))) void admpc_fused20_prepare(void)
{
    // 18.2 KB per workgroup: below the 64 KB default, no opt-in needed; kept for symmetry with the other units
}

// sched ints of a handle that solves up to `cap` instances per call: TWO scheduler states used alternately (zeroed at allocation; the order kernel of a
// launch zeroes the header of the next launch's state, the last workgroup to leave re-arms its own: no memset in front of a launch -- a hipMemsetAsync
// there cost 2 us per step at configs[1] and 26 us at N = 40)
// one state: header and the bins' lists; in a build with F20_DEFER also the expansion queues (F20_NQ counter lines, cap + F20_NQ entries, padded to an even
// count) and the steps of the pushed expansions (cap x 40 doubles)
__attribute__((visibility("hidden"))) size_t admpc_fused20_state_ints(int cap) {
    size_t ints = (size_t)F20_HDR + (size_t)F20_NB * (size_t)cap;
#if F20_DEFER
    ints += (size_t)F20_NQ * 32 + (size_t)(cap + F20_NQ + (cap & 1)) + 2 * (size_t)40 * (size_t)cap;
#endif
    return ints;
}
__attribute__((visibility("hidden"))) size_t admpc_fused20_sched_ints(int cap) { return 2 * admpc_fused20_state_ints(cap); }      // two states (work_order.h)

// grid: persistent, eight one-wave workgroups per CU (two waves per SIMD); slotbuf: admpc_fused20_slot_doubles(num_cu) doubles
__attribute__((visibility("hidden"))) void admpc_fused20_launch(int num_cu, hipStream_t st, const AdmpcConfig* d_cfg, int B, int qmask,
        const double* x0, const double* yref, const double* yref_e, const double* p, double* xbar, double* ubar,
        double* cost, int32_t* stat, int32_t* iters, int first, int* sched2, int cap, int flip, double* slotbuf)
{
    const int lds = FusedLds::total * (int)sizeof(double);
    int grid = num_cu * 8; if (grid > B) grid = B;
    // every launch pair is self-contained: ticket counter, exit counter and bin counts start from zero on the caller's stream (the last
    // workgroup to leave re-arms them as well; a launch that failed half-way, or a handle misused from two streams, cannot poison the next)
    const size_t one = admpc_fused20_state_ints(cap);
    int* const sched = sched2 + (flip ? one : 0);
    int* const sched_next = sched2 + (flip ? 0 : one);
    const int kcap = grid == B ? 0 : cap;      // the batch fits the grid: no work order (work_order.h: f20_next)
    if (kcap) hipLaunchKernelGGL(admpc_f20_order_kernel, dim3((B + 255) / 256), dim3(256), 0, st, d_cfg, B, x0, yref, yref_e, sched, cap, sched_next);
    if (qmask == 7)
        hipLaunchKernelGGL((admpc_fused20_kernel<7>), dim3(grid), dim3(WAVE), lds, st, d_cfg, B, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched, kcap, slotbuf);
    else
        hipLaunchKernelGGL((admpc_fused20_kernel<127>), dim3(grid), dim3(WAVE), lds, st, d_cfg, B, x0, yref, yref_e, p, xbar, ubar, cost, stat, iters, first, sched, kcap, slotbuf);
}

}
