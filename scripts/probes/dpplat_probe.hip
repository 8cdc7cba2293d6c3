// Latency / issue probe for the DPP FMAC forms kernel R is built from (lone wave, s_memtime ticks per instruction).
//   A: 64 dependent v_fma_f64            B: 64 dependent v_fmac_f64_dpp (chain through the accumulator, DPP source independent)
//   C: 64 v_fmac_f64_dpp round-robin into 8 accumulators     D: 64 independent v_fma_f64 (8 accumulators)
//   E: 64 dependent v_fmac_f64_dpp whose DPP source is the accumulator itself (s_nop 1 in between, as the hazard demands)
//   F..H: A..C in fp32     I: 64 dependent v_add_f64    J: dependent v_mov_b64_dpp + v_fma_f64 pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory")
#define T1(slot) asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); if (threadIdx.x == 0) out[slot] = t1 - t0
__global__ void k(unsigned long long* out, double* sink, double a0, double b0, float fa0, float fb0)
{
    unsigned long long t0, t1;
    double acc[8], a = a0 + threadIdx.x * 1e-9, b = b0;
    float facc[8], fa = fa0 + threadIdx.x * 1e-6f, fb = fb0;
    for (int i = 0; i < 8; ++i) { acc[i] = i * a; facc[i] = i * fa; }
    for (int rep = 0; rep < 3; ++rep) {          // the last repetition counts (instruction cache warm)
        T0();
        REP8(REP8(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));))
        T1(0);
        T0();
        REP8(REP8(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[0]) : "v"(a), "v"(b));))
        T1(1);
        T0();
        REP8(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                          "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                          "v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                          "v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : "v"(a), "v"(b));)
        T1(2);
        T0();
        REP8(asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                          "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7"
                          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : "v"(a), "v"(b));)
        T1(3);
        T0();
        REP8(REP8(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[1]) : "v"(b));))
        T1(4);
        T0();
        REP8(REP8(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(facc[0]) : "v"(fa), "v"(fb));))
        T1(5);
        T0();
        REP8(REP8(asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(facc[0]) : "v"(fa), "v"(fb));))
        T1(6);
        T0();
        REP8(asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                          "v_fmac_f32_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                          "v_fmac_f32_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                          "v_fmac_f32_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                          : "+v"(facc[0]), "+v"(facc[1]), "+v"(facc[2]), "+v"(facc[3]), "+v"(facc[4]), "+v"(facc[5]), "+v"(facc[6]), "+v"(facc[7]) : "v"(fa), "v"(fb));)
        T1(7);
        T0();
        REP8(REP8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[2]) : "v"(b));))
        T1(8);
        T0();
        REP8(REP8(asm volatile("s_nop 1\n\tv_mov_b32_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fma_f32 %0, %1, %2, %0" : "+v"(facc[1]), "=&v"(facc[2]) : "v"(fb));))
        T1(9);
        T0();
        REP8(REP8(asm volatile("s_nop 0" ::);))
        T1(10);
    }
    double s = 0; float fs = 0;
    for (int i = 0; i < 8; ++i) { s += acc[i]; fs += facc[i]; }
    sink[threadIdx.x] = s + fs;
}
int main()
{
    unsigned long long* d; double* sink; hipMalloc(&d, 16 * 8); hipMalloc(&sink, 64 * 8);
    k<<<1, 64>>>(d, sink, 1.0000001, 0.9999999, 1.0001f, 0.9999f);
    unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[11] = {"A dependent v_fma_f64", "B dependent v_fmac_f64_dpp (chain through acc)", "C v_fmac_f64_dpp, 8 accumulators", "D v_fma_f64, 8 accumulators",
                          "E dependent v_fmac_f64_dpp, DPP source = acc (+ s_nop 1)", "F dependent v_fma_f32", "G dependent v_fmac_f32_dpp", "H v_fmac_f32_dpp, 8 accumulators",
                          "I dependent v_add_f64", "J dependent (s_nop 1 + v_mov_b32_dpp + v_fma_f32)", "K s_nop 0 (overhead reference)"};
    for (int i = 0; i < 11; ++i) printf("%-58s %6llu ticks / 64 = %6.2f per instruction\n", nm[i], h[i], h[i] / 64.0);
    return 0;
}
