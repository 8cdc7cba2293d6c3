// rowqp_dev.h -- gfx950 backend of rowqp_core.h: one MPC instance per 16-lane DPP row, four per wavefront.
//
// Every exchange between the lanes of an instance is a DPP row operation:
//   acc += lane_n(src) * coef        v_fmac_f64_dpp / v_fmac_f32_dpp  acc, src, coef  row_newbcast:n      (ONE instruction)
//   lane_n(v)                        v_mov_b64_dpp / v_mov_b32_dpp    row_newbcast:n
//   neighbour / butterfly steps      v_mov_b32_dpp quad_perm / row_half_mirror / row_mirror  (reductions inside a row)
// The matrix blocks (P*G, G'*M, the Schur update, the mat-vecs) are single inline-assembly statements of 7 .. 51 such
// instructions.  Hazards: hipcc's hazard recogniser does not look inside inline assembly, and a DPP read of a VGPR needs two
// wait states after the VALU write of that register -> every statement starts with `s_nop 1` (covers a producer in compiler
// code), never reads through DPP a register it wrote fewer than two instructions earlier, and its results are consumed by
// compiler code only through ordinary (non-DPP) operands or through the statements below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RQ_FN __device__ __forceinline__
#define RQ_UNROLL _Pragma("unroll")
#define RQ_NOUNROLL _Pragma("unroll 1")
#include "rowqp_core.h"

#define RQ_DPPM " row_mask:0xf bank_mask:0xf\n\t"

template <class T_> struct DevX;

// optional in-kernel phase timers (-DADMPC_PHASE_TIMERS): s_memtime deltas between the stamps of RowQp::solve, summed over all waves
#ifdef ADMPC_PHASE_TIMERS
__device__ unsigned long long g_rq_ticks[16];
__device__ unsigned long long g_rq_last[8192];
#endif

// ---------------------------------------------------------------------------------------------------------------------
// common part
// ---------------------------------------------------------------------------------------------------------------------
template <class T_, class Derived>
struct DevCommon {
    typedef T_ T;
    typedef T V;
    typedef int I;
    typedef bool M;
    typedef __attribute__((address_space(3))) T LT;
    struct Lds { LT* base; bool ok; };                    // ok = false: a row without an LDS region of its own (loads alias row 0, stores are dropped)

    RQ_FN static V splat(T x) { return x; }
    RQ_FN static I isplat(int x) { return x; }
    RQ_FN static I lane() { return (int)(threadIdx.x & 15u); }
    RQ_FN static M mtrue() { return true; }
    RQ_FN static M mfalse() { return false; }
    RQ_FN static M mfrom(bool b) { return b; }
    RQ_FN static V sel(M m, V a, V b) { return m ? a : b; }
    RQ_FN static I isel(M m, I a, I b) { return m ? a : b; }
    RQ_FN static V vabs(V a) { return __builtin_fabs(a); }
    RQ_FN static V vmaxnan(V a, V b) { return (b > a || b != b) ? b : a; }
    // memory
    // Masked stores are unconditional stores whose address is redirected to a dump slot of the lane (the 16-value header of the row's LDS region) -- one v_cndmask instead of an EXEC save / branch / restore per store.
    RQ_FN static V lds_ld(const Lds& L, I off, int imm) { return L.base[off + imm]; }
    RQ_FN static void lds_st(const Lds& L, I off, int imm, V v, M m) { L.base[(m && L.ok) ? off + imm : (int)(threadIdx.x & 15u)] = v; }
    RQ_FN static void lds_ld2(const Lds& L, I off, int imm, V& a, V& b) { a = L.base[off + imm]; b = L.base[off + imm + 1]; }
    RQ_FN static void lds_st2(const Lds& L, I off, int imm, V a, V b, M m) {
        const int o = (m && L.ok) ? off + imm : (int)(threadIdx.x & 14u);
        L.base[o] = a; L.base[o + 1] = b;
    }
    // orders what one sweep / pass wrote (LDS records, workspace in global memory) before what the next one reads.  One wave per
    // workgroup: the wave's own memory counters order the accesses; the workspace lines the CU's vector L1 still holds from an
    // earlier sweep are dropped (acquire at agent scope = buffer_inv sc1) -- the L1 does not follow the wave's own stores
    // (measured: stale states after the roll-out, run-to-run different results).  No sweep reads what it wrote itself.
    RQ_FN static void fence() {
#ifdef ADMPC_PHASE_TIMERS
        unsigned long long f0, f1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(f0) :: "memory");
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifdef ADMPC_PHASE_TIMERS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(f1) :: "memory");
        if (threadIdx.x == 0) { atomicAdd(&g_rq_ticks[14], f1 - f0); atomicAdd(&g_rq_ticks[15], 1ull); }      // slots outside the phase sum
#endif
    }
    // scheduling fence: memory operations are not moved across it, and what is computed from a, b, c, d starts after it (hipcc otherwise
    // sinks the loads of the next step below the arithmetic of the current one and waits for them at once)
    RQ_FN static void sched_fence(V& a, V& b, V& c, V& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory"); }
    // global arrays: uniform base + 32-bit byte offset of the lane (global_load ... v_off, s[base:base+1]); the host splits
    // batches whose arrays would exceed 4 GB
    RQ_FN static const T* gaddr(const T* p, I off) { return (const T*)((const char*)p + (unsigned)off * (unsigned)sizeof(T)); }
    RQ_FN static V gld(const T* p, I off) { return *gaddr(p, off); }
    RQ_FN static void gst(T* p, I off, V v, M m) { if (m) *(T*)((char*)p + (unsigned)off * (unsigned)sizeof(T)) = v; }
    // workspace accesses: written by one sweep, read by a later one of the same wave (fence() in between)
    RQ_FN static V wld(const T* p, I off) { return *gaddr(p, off); }
    // masked stores are unconditional stores whose address is redirected to a dump slot (see lds_st): an EXEC branch around a store
    // makes hipcc wait for every outstanding load at the join
    RQ_FN static void wst(T* p, I off, I dump, V v, M m) { *(T*)((char*)p + (unsigned)(m ? off : dump) * (unsigned)sizeof(T)) = v; }
    // pairs (even offsets): one 16-byte (fp64) / 8-byte (fp32) access
    typedef T T2 __attribute__((ext_vector_type(2)));
    RQ_FN static void wld2(const T* p, I off, V& a, V& b) { const T2 v = *(const T2*)gaddr(p, off); a = v.x; b = v.y; }
    RQ_FN static void wst2(T* p, I off, I dump, V a, V b, M m) { T2 v; v.x = a; v.y = b; *(T2*)((char*)p + (unsigned)(m ? off : dump) * (unsigned)sizeof(T)) = v; }
    // row-uniform logic through the wave ballot
    RQ_FN static unsigned rowbits(M m) {
        const unsigned long long b = __ballot(m);
        return (unsigned)(b >> (threadIdx.x & 48u)) & 0xffffu;
    }
    RQ_FN static M row_and(M m) { return rowbits(m) == 0xffffu; }
    RQ_FN static M row_or(M m) { return rowbits(m) != 0u; }
    RQ_FN static bool any(M m) { return __any(m) != 0; }
#ifdef ADMPC_PHASE_TIMERS
    // the interval that ENDS at stamp(id) is charged to phase id
    RQ_FN static void stamp(int id) {
        unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
        if (threadIdx.x == 0) { if (id != 0) atomicAdd(&g_rq_ticks[id], t - g_rq_last[blockIdx.x]); g_rq_last[blockIdx.x] = t; }
    }
#else
    RQ_FN static void stamp(int) {}
#endif
};

// ---------------------------------------------------------------------------------------------------------------------
// fp64
// ---------------------------------------------------------------------------------------------------------------------
template <>
struct DevX<double> : DevCommon<double, DevX<double>> {
    RQ_FN static V fma(V a, V b, V c) { return __builtin_fma(a, b, c); }
    RQ_FN static V vmax(V a, V b) { return __builtin_fmax(a, b); }
    RQ_FN static V vmin(V a, V b) { return __builtin_fmin(a, b); }
    // 1/d: hardware estimate + two Newton steps (full double accuracy, no range handling: arguments are slacks, multipliers, pivots)
    RQ_FN static V rcp(V d) {
        double r = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-d, r, 1.0);
        return __builtin_fma(r, e, r);
    }
    RQ_FN static void gld6(const T* p, I off, V out[6]) {
        const double2* q2 = reinterpret_cast<const double2*>(gaddr(p, off));
        const double2 a = q2[0], b = q2[1], c = q2[2];
        out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y; out[4] = c.x; out[5] = c.y;
    }
    template <int CTRL> RQ_FN static V dppmov(V v) {              // compiler-visible 32-bit DPP moves (it pads their hazards itself)
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    RQ_FN static V swap1(V v) { return dppmov<0xB1>(v); }        // quad_perm:[1,0,3,2]
    template <class F> RQ_FN static V row_red(V v, F f) {         // butterfly: every lane of the row ends with the row's reduction
        v = f(v, dppmov<0xB1>(v)); v = f(v, dppmov<0x4E>(v));     // quad_perm [1,0,3,2], [2,3,0,1]
        v = f(v, dppmov<0x141>(v)); v = f(v, dppmov<0x140>(v));   // row_half_mirror, row_mirror
        return v;
    }
    RQ_FN static V row_sum(V v) { return row_red(v, [](V a, V b) { return a + b; }); }
    RQ_FN static V row_max(V v) { return row_red(v, [](V a, V b) { return __builtin_fmax(a, b); }); }
    RQ_FN static V row_maxnan(V v) { return row_red(v, [](V a, V b) { return (b > a || b != b) ? b : a; }); }

    template <int L> RQ_FN static V bc(V v) {
        V r;
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2" RQ_DPPM : "=v"(r) : "v"(v), "n"(L));
        return r;
    }
#define RQ_F64(acc, src, co, ln) "v_fmac_f64_dpp %[" #acc "], %[" #src "], %[" #co "] row_newbcast:" #ln RQ_DPPM
    // acc += sum_{i<7} lane_{L0+i}(src) * coef[i]
    template <int n, int L0> RQ_FN static void dotbc(const V* c, V src, V& acc) {
        static_assert(n == 7 && (L0 == 0 || L0 == 2), "dotbc: 7 terms from lane 0 or 2");
        if constexpr (L0 == 0)
            asm volatile("s_nop 1\n\t" RQ_F64(a, s, c0, 0) RQ_F64(a, s, c1, 1) RQ_F64(a, s, c2, 2) RQ_F64(a, s, c3, 3) RQ_F64(a, s, c4, 4) RQ_F64(a, s, c5, 5) RQ_F64(a, s, c6, 6)
                         : [a] "+v"(acc) : [s] "v"(src), [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]));
        else
            asm volatile("s_nop 1\n\t" RQ_F64(a, s, c0, 2) RQ_F64(a, s, c1, 3) RQ_F64(a, s, c2, 4) RQ_F64(a, s, c3, 5) RQ_F64(a, s, c4, 6) RQ_F64(a, s, c5, 7) RQ_F64(a, s, c6, 8)
                         : [a] "+v"(acc) : [s] "v"(src), [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]));
    }
    // acc += sum_{c<7} lane_c(w0) * e0 + lane_c(w1) * e1
    RQ_FN static void sumbc2(V w0, V w1, V e0, V e1, V& acc) {
        asm volatile("s_nop 1\n\t" RQ_F64(a, u, p, 0) RQ_F64(a, v, q, 0) RQ_F64(a, u, p, 1) RQ_F64(a, v, q, 1) RQ_F64(a, u, p, 2) RQ_F64(a, v, q, 2) RQ_F64(a, u, p, 3) RQ_F64(a, v, q, 3)
                     RQ_F64(a, u, p, 4) RQ_F64(a, v, q, 4) RQ_F64(a, u, p, 5) RQ_F64(a, v, q, 5) RQ_F64(a, u, p, 6) RQ_F64(a, v, q, 6)
                     : [a] "+v"(acc) : [u] "v"(w0), [v] "v"(w1), [p] "v"(e0), [q] "v"(e1));
    }
#define RQ_PG_ROW64(i) "v_mov_b64 %[m" #i "], 0\n\t" RQ_F64(m##i, p##i, g0, 0) RQ_F64(m##i, p##i, g1, 1) RQ_F64(m##i, p##i, g2, 2) RQ_F64(m##i, p##i, g3, 3) \
                       RQ_F64(m##i, p##i, g4, 4) RQ_F64(m##i, p##i, g5, 5) RQ_F64(m##i, p##i, g6, 6)
    // M[i] = sum_l lane_l(P[i]) * G[l]   (P symmetric: lane l holds column l)
    RQ_FN static void pg(const V P[7], const V G[7], V Mm[7]) {
        asm volatile("s_nop 1\n\t" RQ_PG_ROW64(0) RQ_PG_ROW64(1) RQ_PG_ROW64(2) RQ_PG_ROW64(3) RQ_PG_ROW64(4) RQ_PG_ROW64(5) RQ_PG_ROW64(6)
                     : [m0] "=&v"(Mm[0]), [m1] "=&v"(Mm[1]), [m2] "=&v"(Mm[2]), [m3] "=&v"(Mm[3]), [m4] "=&v"(Mm[4]), [m5] "=&v"(Mm[5]), [m6] "=&v"(Mm[6])
                     : [p0] "v"(P[0]), [p1] "v"(P[1]), [p2] "v"(P[2]), [p3] "v"(P[3]), [p4] "v"(P[4]), [p5] "v"(P[5]), [p6] "v"(P[6]),
                       [g0] "v"(G[0]), [g1] "v"(G[1]), [g2] "v"(G[2]), [g3] "v"(G[3]), [g4] "v"(G[4]), [g5] "v"(G[5]), [g6] "v"(G[6]));
    }
#define RQ_GTM_ROW64(r) "v_mov_b64 %[h" #r "], 0\n\t" RQ_F64(h##r, g0, m0, r) RQ_F64(h##r, g1, m1, r) RQ_F64(h##r, g2, m2, r) RQ_F64(h##r, g3, m3, r) \
                        RQ_F64(h##r, g4, m4, r) RQ_F64(h##r, g5, m5, r)
    // H[r] = sum_{l<6} lane_r(G[l]) * M[l] for r = 2..8, plus the structural row 6 of [A B] (delta' = delta + h u1); H[0..1] = M[0..1]
    RQ_FN static void gtm(const V G[6], const V Mm[7], T h, V H[9]) {
        H[0] = Mm[0]; H[1] = Mm[1];
        asm volatile("s_nop 1\n\t" RQ_GTM_ROW64(2) RQ_GTM_ROW64(3) RQ_GTM_ROW64(4) RQ_GTM_ROW64(5) RQ_GTM_ROW64(6) RQ_GTM_ROW64(7) RQ_GTM_ROW64(8)
                     "v_add_f64 %[h6], %[h6], %[m6]\n\tv_fmac_f64 %[h8], %[hh], %[m6]"
                     : [h2] "=&v"(H[2]), [h3] "=&v"(H[3]), [h4] "=&v"(H[4]), [h5] "=&v"(H[5]), [h6] "=&v"(H[6]), [h7] "=&v"(H[7]), [h8] "=&v"(H[8])
                     : [g0] "v"(G[0]), [g1] "v"(G[1]), [g2] "v"(G[2]), [g3] "v"(G[3]), [g4] "v"(G[4]), [g5] "v"(G[5]),
                       [m0] "v"(Mm[0]), [m1] "v"(Mm[1]), [m2] "v"(Mm[2]), [m3] "v"(Mm[3]), [m4] "v"(Mm[4]), [m5] "v"(Mm[5]), [m6] "v"(Mm[6]), [hh] "v"(h));
    }
    // H[i] += lane_7(H[i]) * K0 + lane_8(H[i]) * K1 for i < 7 (K0 = K1 = 0 on lanes 7, 8, so their H[i] stay the broadcast sources)
    RQ_FN static void schur(V H[9], V K0, V K1) {
        asm volatile("s_nop 1\n\t" RQ_F64(h0, h0, k0, 7) RQ_F64(h1, h1, k0, 7) RQ_F64(h2, h2, k0, 7) RQ_F64(h3, h3, k0, 7) RQ_F64(h4, h4, k0, 7) RQ_F64(h5, h5, k0, 7) RQ_F64(h6, h6, k0, 7)
                     RQ_F64(h0, h0, k1, 8) RQ_F64(h1, h1, k1, 8) RQ_F64(h2, h2, k1, 8) RQ_F64(h3, h3, k1, 8) RQ_F64(h4, h4, k1, 8) RQ_F64(h5, h5, k1, 8) RQ_F64(h6, h6, k1, 8)
                     : [h0] "+v"(H[0]), [h1] "+v"(H[1]), [h2] "+v"(H[2]), [h3] "+v"(H[3]), [h4] "+v"(H[4]), [h5] "+v"(H[5]), [h6] "+v"(H[6])
                     : [k0] "v"(K0), [k1] "v"(K1));
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// fp32
// ---------------------------------------------------------------------------------------------------------------------
template <>
struct DevX<float> : DevCommon<float, DevX<float>> {
    RQ_FN static V fma(V a, V b, V c) { return __builtin_fmaf(a, b, c); }
    RQ_FN static V vmax(V a, V b) { return __builtin_fmaxf(a, b); }
    RQ_FN static V vmin(V a, V b) { return __builtin_fminf(a, b); }
    RQ_FN static V rcp(V d) {                                      // v_rcp_f32 (1 ulp) + one Newton step
        const float r = __builtin_amdgcn_rcpf(d);
        return __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
    }
    RQ_FN static void gld6(const T* p, I off, V out[6]) {
        const float2* q2 = reinterpret_cast<const float2*>(gaddr(p, off));
        const float2 a = q2[0], b = q2[1], c = q2[2];
        out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y; out[4] = c.x; out[5] = c.y;
    }
    template <int CTRL> RQ_FN static V dppmov(V v) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
    }
    RQ_FN static V swap1(V v) { return dppmov<0xB1>(v); }
    template <class F> RQ_FN static V row_red(V v, F f) {
        v = f(v, dppmov<0xB1>(v)); v = f(v, dppmov<0x4E>(v));
        v = f(v, dppmov<0x141>(v)); v = f(v, dppmov<0x140>(v));
        return v;
    }
    RQ_FN static V row_sum(V v) { return row_red(v, [](V a, V b) { return a + b; }); }
    RQ_FN static V row_max(V v) { return row_red(v, [](V a, V b) { return __builtin_fmaxf(a, b); }); }
    RQ_FN static V row_maxnan(V v) { return row_red(v, [](V a, V b) { return (b > a || b != b) ? b : a; }); }

    template <int L> RQ_FN static V bc(V v) {
        V r;
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2" RQ_DPPM : "=v"(r) : "v"(v), "n"(L));
        return r;
    }
#define RQ_F32(acc, src, co, ln) "v_fmac_f32_dpp %[" #acc "], %[" #src "], %[" #co "] row_newbcast:" #ln RQ_DPPM
    template <int n, int L0> RQ_FN static void dotbc(const V* c, V src, V& acc) {
        static_assert(n == 7 && (L0 == 0 || L0 == 2), "dotbc: 7 terms from lane 0 or 2");
        if constexpr (L0 == 0)
            asm volatile("s_nop 1\n\t" RQ_F32(a, s, c0, 0) RQ_F32(a, s, c1, 1) RQ_F32(a, s, c2, 2) RQ_F32(a, s, c3, 3) RQ_F32(a, s, c4, 4) RQ_F32(a, s, c5, 5) RQ_F32(a, s, c6, 6)
                         : [a] "+v"(acc) : [s] "v"(src), [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]));
        else
            asm volatile("s_nop 1\n\t" RQ_F32(a, s, c0, 2) RQ_F32(a, s, c1, 3) RQ_F32(a, s, c2, 4) RQ_F32(a, s, c3, 5) RQ_F32(a, s, c4, 6) RQ_F32(a, s, c5, 7) RQ_F32(a, s, c6, 8)
                         : [a] "+v"(acc) : [s] "v"(src), [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]));
    }
    RQ_FN static void sumbc2(V w0, V w1, V e0, V e1, V& acc) {
        asm volatile("s_nop 1\n\t" RQ_F32(a, u, p, 0) RQ_F32(a, v, q, 0) RQ_F32(a, u, p, 1) RQ_F32(a, v, q, 1) RQ_F32(a, u, p, 2) RQ_F32(a, v, q, 2) RQ_F32(a, u, p, 3) RQ_F32(a, v, q, 3)
                     RQ_F32(a, u, p, 4) RQ_F32(a, v, q, 4) RQ_F32(a, u, p, 5) RQ_F32(a, v, q, 5) RQ_F32(a, u, p, 6) RQ_F32(a, v, q, 6)
                     : [a] "+v"(acc) : [u] "v"(w0), [v] "v"(w1), [p] "v"(e0), [q] "v"(e1));
    }
#define RQ_PG_ROW32(i) "v_mov_b32 %[m" #i "], 0\n\t" RQ_F32(m##i, p##i, g0, 0) RQ_F32(m##i, p##i, g1, 1) RQ_F32(m##i, p##i, g2, 2) RQ_F32(m##i, p##i, g3, 3) \
                       RQ_F32(m##i, p##i, g4, 4) RQ_F32(m##i, p##i, g5, 5) RQ_F32(m##i, p##i, g6, 6)
    RQ_FN static void pg(const V P[7], const V G[7], V Mm[7]) {
        asm volatile("s_nop 1\n\t" RQ_PG_ROW32(0) RQ_PG_ROW32(1) RQ_PG_ROW32(2) RQ_PG_ROW32(3) RQ_PG_ROW32(4) RQ_PG_ROW32(5) RQ_PG_ROW32(6)
                     : [m0] "=&v"(Mm[0]), [m1] "=&v"(Mm[1]), [m2] "=&v"(Mm[2]), [m3] "=&v"(Mm[3]), [m4] "=&v"(Mm[4]), [m5] "=&v"(Mm[5]), [m6] "=&v"(Mm[6])
                     : [p0] "v"(P[0]), [p1] "v"(P[1]), [p2] "v"(P[2]), [p3] "v"(P[3]), [p4] "v"(P[4]), [p5] "v"(P[5]), [p6] "v"(P[6]),
                       [g0] "v"(G[0]), [g1] "v"(G[1]), [g2] "v"(G[2]), [g3] "v"(G[3]), [g4] "v"(G[4]), [g5] "v"(G[5]), [g6] "v"(G[6]));
    }
#define RQ_GTM_ROW32(r) "v_mov_b32 %[h" #r "], 0\n\t" RQ_F32(h##r, g0, m0, r) RQ_F32(h##r, g1, m1, r) RQ_F32(h##r, g2, m2, r) RQ_F32(h##r, g3, m3, r) \
                        RQ_F32(h##r, g4, m4, r) RQ_F32(h##r, g5, m5, r)
    RQ_FN static void gtm(const V G[6], const V Mm[7], T h, V H[9]) {
        H[0] = Mm[0]; H[1] = Mm[1];
        asm volatile("s_nop 1\n\t" RQ_GTM_ROW32(2) RQ_GTM_ROW32(3) RQ_GTM_ROW32(4) RQ_GTM_ROW32(5) RQ_GTM_ROW32(6) RQ_GTM_ROW32(7) RQ_GTM_ROW32(8)
                     "v_add_f32 %[h6], %[h6], %[m6]\n\tv_fmac_f32 %[h8], %[hh], %[m6]"
                     : [h2] "=&v"(H[2]), [h3] "=&v"(H[3]), [h4] "=&v"(H[4]), [h5] "=&v"(H[5]), [h6] "=&v"(H[6]), [h7] "=&v"(H[7]), [h8] "=&v"(H[8])
                     : [g0] "v"(G[0]), [g1] "v"(G[1]), [g2] "v"(G[2]), [g3] "v"(G[3]), [g4] "v"(G[4]), [g5] "v"(G[5]),
                       [m0] "v"(Mm[0]), [m1] "v"(Mm[1]), [m2] "v"(Mm[2]), [m3] "v"(Mm[3]), [m4] "v"(Mm[4]), [m5] "v"(Mm[5]), [m6] "v"(Mm[6]), [hh] "v"(h));
    }
    RQ_FN static void schur(V H[9], V K0, V K1) {
        asm volatile("s_nop 1\n\t" RQ_F32(h0, h0, k0, 7) RQ_F32(h1, h1, k0, 7) RQ_F32(h2, h2, k0, 7) RQ_F32(h3, h3, k0, 7) RQ_F32(h4, h4, k0, 7) RQ_F32(h5, h5, k0, 7) RQ_F32(h6, h6, k0, 7)
                     RQ_F32(h0, h0, k1, 8) RQ_F32(h1, h1, k1, 8) RQ_F32(h2, h2, k1, 8) RQ_F32(h3, h3, k1, 8) RQ_F32(h4, h4, k1, 8) RQ_F32(h5, h5, k1, 8) RQ_F32(h6, h6, k1, 8)
                     : [h0] "+v"(H[0]), [h1] "+v"(H[1]), [h2] "+v"(H[2]), [h3] "+v"(H[3]), [h4] "+v"(H[4]), [h5] "+v"(H[5]), [h6] "+v"(H[6])
                     : [k0] "v"(K0), [k1] "v"(K1));
    }
};
