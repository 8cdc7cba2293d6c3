// Micro-probe: issue cost / dependent latency of the fp64 vector instructions the interior-point kernel is made of (gfx950).
// One workgroup of 64 x W threads on one CU (W waves -> W/4 per SIMD when W >= 4); cycles from s_memtime around 64 x 16 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define REP16(x) x x x x x x x x x x x x x x x x
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double m = 1.0000001, c = 1e-9;
    int i0 = threadIdx.x * 8, i1 = i0 + 8, i2 = i0 + 16, i3 = i0 + 24, i4 = i0 * 3, i5 = i0 * 5; i0 &= 0x1f8;
    __shared__ double lds[64]; if (threadIdx.x < 64) lds[threadIdx.x] = 0.0;
    __syncthreads();
    unsigned long long t0 = now();
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {        // 8 independent fma chains
            asm volatile(REP16("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                               "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 1) { // one dependent fma chain
            asm volatile(REP16("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\t"
                               "v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\t")
                         : "+v"(a0) : "v"(m), "v"(c));
        } else if (MODE == 2) { // independent fmac dpp row_newbcast
            asm volatile(REP16("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 3) { // readlane pair -> dependent fma with sgpr operand (the old column broadcast), chain through a0
            asm volatile(REP16("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\tv_fma_f64 %2, s[20:21], %3, %2\n\t"
                               "v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %1, 5\n\tv_fma_f64 %2, s[20:21], %3, %2\n\t"
                               "v_readlane_b32 s20, %0, 7\n\tv_readlane_b32 s21, %1, 7\n\tv_fma_f64 %2, s[20:21], %3, %2\n\t"
                               "v_readlane_b32 s20, %0, 9\n\tv_readlane_b32 s21, %1, 9\n\tv_fma_f64 %2, s[20:21], %3, %2\n\t")
                         : : "v"(__double2loint(a1)), "v"(__double2hiint(a1)), "v"(a0), "v"(c) : "s20", "s21");
        } else if (MODE == 4) { // substitution step: readlane of the accumulator itself (dependent), then fma
            asm volatile(REP16("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\ts_nop 0\n\tv_fma_f64 %2, s[20:21], %3, %2\n\t"
                               "v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %1, 5\n\ts_nop 0\n\tv_fma_f64 %2, s[20:21], %3, %2\n\t")
                         : : "v"(__double2loint(a1)), "v"(__double2hiint(a1)), "v"(a0), "v"(c) : "s20", "s21");
        } else if (MODE == 5) { // rcp chain (dependent)
            asm volatile(REP16("v_rcp_f64 %0, %0\n\tv_rcp_f64 %0, %0\n\tv_rcp_f64 %0, %0\n\tv_rcp_f64 %0, %0\n\t"
                               "v_rcp_f64 %0, %0\n\tv_rcp_f64 %0, %0\n\tv_rcp_f64 %0, %0\n\tv_rcp_f64 %0, %0\n\t") : "+v"(a0));
        } else if (MODE == 6) { // independent rcp
            asm volatile(REP16("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t"
                               "v_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 7) { // 32-bit mov (independent)
            asm volatile(REP16("v_mov_b32 %0, %1\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %2, %3\n\t")
                         : : "v"(__double2loint(a0)), "v"(__double2hiint(a0)), "v"(__double2loint(a1)), "v"(__double2hiint(a1)));
        } else if (MODE == 8) { // fp64 mul/add mix independent (v_mul_f64, v_add_f64)
            asm volatile(REP16("v_mul_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %9\n\tv_mul_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %9\n\t"
                               "v_mul_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %9\n\tv_mul_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 9) { // v_cndmask_b32 independent
            asm volatile(REP16("v_cndmask_b32 %0, %1, %2, vcc\n\tv_cndmask_b32 %3, %1, %2, vcc\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cndmask_b32 %3, %1, %2, vcc\n\t"
                               "v_cndmask_b32 %0, %1, %2, vcc\n\tv_cndmask_b32 %3, %1, %2, vcc\n\tv_cndmask_b32 %0, %1, %2, vcc\n\tv_cndmask_b32 %3, %1, %2, vcc\n\t")
                         : : "v"(__double2loint(a0)), "v"(__double2hiint(a0)), "v"(__double2loint(a1)), "v"(__double2hiint(a1)) : "vcc");
        } else if (MODE == 10) { // v_cndmask_b32 with an SGPR-pair mask (VOP3), distinct destinations
            asm volatile("s_mov_b64 s[20:21], 0x5555\n\t" REP16("v_cndmask_b32 %0, %4, %5, s[20:21]\n\tv_cndmask_b32 %1, %4, %5, s[20:21]\n\tv_cndmask_b32 %2, %4, %5, s[20:21]\n\tv_cndmask_b32 %3, %4, %5, s[20:21]\n\t"
                               "v_cndmask_b32 %0, %5, %4, s[20:21]\n\tv_cndmask_b32 %1, %5, %4, s[20:21]\n\tv_cndmask_b32 %2, %5, %4, s[20:21]\n\tv_cndmask_b32 %3, %5, %4, s[20:21]\n\t")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4), "v"(i5) : "s20", "s21");
        } else if (MODE == 11) { // v_cmp_gt_f64 -> sgpr pair, independent
            asm volatile(REP16("v_cmp_gt_f64 s[20:21], %0, %1\n\tv_cmp_gt_f64 s[22:23], %1, %0\n\tv_cmp_gt_f64 s[20:21], %0, %1\n\tv_cmp_gt_f64 s[22:23], %1, %0\n\t"
                               "v_cmp_gt_f64 s[20:21], %0, %1\n\tv_cmp_gt_f64 s[22:23], %1, %0\n\tv_cmp_gt_f64 s[20:21], %0, %1\n\tv_cmp_gt_f64 s[22:23], %1, %0\n\t")
                         : : "v"(a0), "v"(a1) : "s20", "s21", "s22", "s23");
        } else if (MODE == 12) { // compare + two selects (a double select as the compiler writes it)
            asm volatile(REP16("v_cmp_gt_i32 vcc, %4, %5\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc\n\tv_cmp_gt_i32 vcc, %5, %4\n\tv_cndmask_b32 %2, %4, %5, vcc\n\tv_cndmask_b32 %3, %5, %4, vcc\n\t"
                               "v_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %3\n\t")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4), "v"(i5) : "vcc");
        } else if (MODE == 13) { // permlane32 swap
            asm volatile(REP16("v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1\n\t")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
        } else if (MODE == 14) { // ds_read_b64 dependent chain (address from data)
            asm volatile(REP16("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0x1f8, %2\n\t" "ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0x1f8, %2\n\t"
                               "ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0x1f8, %2\n\t" "ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0x1f8, %2\n\t")
                         : "+v"(a0), "+v"(i0) : "v"(i1));
        } else if (MODE == 15) { // v_readlane pairs independent (sgpr writes)
            asm volatile(REP16("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\tv_readlane_b32 s22, %0, 5\n\tv_readlane_b32 s23, %1, 5\n\t"
                               "v_readlane_b32 s20, %0, 7\n\tv_readlane_b32 s21, %1, 7\n\tv_readlane_b32 s22, %0, 9\n\tv_readlane_b32 s23, %1, 9\n\t")
                         : : "v"(i0), "v"(i1) : "s20", "s21", "s22", "s23");
        }
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + lds[threadIdx.x & 63];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE> void run(const char* name, int ninstr_per_it, double* out, unsigned long long* cyc) {
    for (int waves : {1, 4, 8, 16}) {       // waves per workgroup on one CU: 4 -> 1 per SIMD, 8 -> 2 per SIMD
        k<MODE><<<1, 64 * waves>>>(out, cyc, 1.0);
        unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-34s waves/CU %2d: %.2f s_memtime ticks per instruction per wave\n", name, waves, mx / (64.0 * ninstr_per_it));
    }
}
int main() {
    double* out; unsigned long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8 * 64);
    run<0>("fma_f64 x8 independent", 128, out, cyc);
    run<1>("fma_f64 dependent chain", 128, out, cyc);
    run<2>("fmac_f64_dpp row_newbcast indep", 128, out, cyc);
    run<3>("2 readlane + fma(sgpr) acc chain", 192, out, cyc);
    run<4>("2 readlane + nop + fma", 128, out, cyc);
    run<5>("rcp_f64 dependent", 128, out, cyc);
    run<6>("rcp_f64 independent", 128, out, cyc);
    run<7>("v_mov_b32 independent", 128, out, cyc);
    run<8>("mul/add f64 independent", 128, out, cyc);
    run<9>("v_cndmask_b32 independent", 128, out, cyc);
    run<10>("v_cndmask_b32 sgpr mask", 128, out, cyc);
    run<11>("v_cmp_gt_f64 -> sgpr", 128, out, cyc);
    run<12>("cmp + 2 cndmask (+add)", 128, out, cyc);
    run<13>("permlane swap + s_nop 1", 64, out, cyc);
    run<14>("ds_read_b64 dependent", 64, out, cyc);
    run<15>("v_readlane_b32 independent", 128, out, cyc);
    return 0;
}
