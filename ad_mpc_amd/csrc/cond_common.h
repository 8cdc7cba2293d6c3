// cond_common.h -- helpers shared by the condensed N = 20 pipeline (admpc_kernels.hip: kernels C, D, E) and its fused successor
// (admpc_fused20.hip): lane-scan primitives, HBM <-> LDS staging of one instance, the LDS map.  Include inside the translation
// unit's anonymous namespace after dense40.h (dpp_mov, WAVE) and after GTS / NX are defined.
#pragma once

// x / 7 for 0 <= x < 13107 as a 32-bit multiply-shift: hipcc 7.2 narrows small non-negative ints to 16 bits and its backend
// cannot select the 16-bit udivrem by 7 at -Oz ("Cannot select: i16 udivrem"); there is no `/ 7` or `% 7` on the device.
__device__ __forceinline__ int div7(int x) { return (int)(((unsigned)x * 9363u) >> 16); }


template <class Op>
__device__ __forceinline__ double wave_scan_incl(double v) {      // inclusive prefix over lanes 0..lane
    v = Op::f(v, dpp_shr<0x111, Op::zf>(Op::id(), v));
    v = Op::f(v, dpp_shr<0x112, Op::zf>(Op::id(), v));
    v = Op::f(v, dpp_shr<0x114, Op::zf>(Op::id(), v));
    v = Op::f(v, dpp_shr<0x118, Op::zf>(Op::id(), v));
    v = Op::f(v, dpp_mov<0x142, 0xa>(Op::id(), v));
    v = Op::f(v, dpp_mov<0x143, 0xc>(Op::id(), v));
    return v;
}

// 1/sqrt(d) for well-scaled positive d: hardware estimate + two Newton steps (full double accuracy for the pivots seen here,
// 1e-3 .. 1e15; ocml's rsqrt adds range scaling that the pivot chain does not need)
__device__ __forceinline__ double rsqrt_nr(double d) {
    double r = __builtin_amdgcn_rsq(d);
    double e = fma(-d * r, r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-d * r, r, 1.0);
    return fma(0.5 * r, e, r);
}

__device__ __forceinline__ int wave_scan_incl_int(int v) {        // inclusive prefix sum over lanes 0..lane
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}


// ---- staging of one instance between HBM and LDS.  All loads of a block are issued before the first use, so a wave pays
//      one memory round trip per block instead of one per 64 elements (the plain copy loop serialises load -> ds_write).
//      CNT doubles, CNT even, both sides 16-byte aligned; the clamped tail re-copies the last element pair (same value).
template <int CNT>
__device__ __forceinline__ void stage_in(double* __restrict__ dst, const double* __restrict__ src, const int lane) {
    static_assert(CNT % 2 == 0, "stage_in copies double2");
    constexpr int C2 = CNT / 2, IT = (C2 + WAVE - 1) / WAVE;
    double2 tmp[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) { int i = lane + WAVE * it; i = i < C2 ? i : C2 - 1; tmp[it] = reinterpret_cast<const double2*>(src)[i]; }
#pragma unroll
    for (int it = 0; it < IT; ++it) { int i = lane + WAVE * it; i = i < C2 ? i : C2 - 1; reinterpret_cast<double2*>(dst)[i] = tmp[it]; }
}
// dq[k][c] = xbar[k][c] - (k < N ? yref[k][c] : yref_e[c]), k = 0..N: the same, for the tracking-error block
template <int NN>
__device__ __forceinline__ void stage_dq(double* __restrict__ dq, const double* __restrict__ xb, const double* __restrict__ yr,
                                         const double* __restrict__ yre, const int lane) {
    constexpr int CNT = (NN + 1) * NX, IT = (CNT + WAVE - 1) / WAVE;
    double xv[IT], yv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        int i = lane + WAVE * it; i = i < CNT ? i : CNT - 1;
        const int k = div7(i), c = i - 7 * k;
        xv[it] = xb[i];
        yv[it] = k < NN ? yr[k * 9 + c] : yre[c];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) { int i = lane + WAVE * it; i = i < CNT ? i : CNT - 1; dq[i] = xv[it] - yv[it]; }
}


template <int NT>
struct DenseLds {
    static constexpr int N = NT, n = 2 * NT, NTRI = n * (n + 1) / 2;
    static constexpr int LSZ = NTRI > N * GTS ? NTRI : N * GTS;
    static constexpr int BLS = (N * 7 + 1) & ~1, DQS = ((N + 1) * 7 + 1) & ~1;     // keep every sub-array 16-byte aligned
    static constexpr int total = 2 * (NTRI + (NTRI & 1)) + 5 * 64 + 4 * 64;     // interior-point kernel: H, L, parked constants, exchange buffers
    static constexpr int expand_total = N * GTS + BLS + DQS + 64;                 // expand kernel: linearisation, defects, tracking error, du
};

