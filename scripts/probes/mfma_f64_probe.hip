// Probe: fp64 rate of the matrix pipe (v_mfma_f64_16x16x4_f64) against the vector pipe (v_fma_f64) on gfx950, whole chip.
// Backs the statement of DESIGN.md that an MFMA variant of the fp64 factorisation has no rate advantage on MI355X (matrix fp64
// peak = vector fp64 peak) and loses on tile utilisation.  Output goes to profiles/r2/mfma_vs_valu_f64.txt.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
#define ITERS 4096

// 8 independent accumulators per lane, 2 flops per lane per instruction
__global__ __launch_bounds__(256) void k_valu(double* out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 0.999999, c = 1e-9;
#pragma unroll 4
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// 4 independent 16x16 accumulator tiles per wave; one v_mfma_f64_16x16x4_f64 = 16*16*4*2 = 2048 flops per wave
__global__ __launch_bounds__(256) void k_mfma(double* out, double seed)
{
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-4;
#pragma unroll 4
    for (int i = 0; i < ITERS; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

static double run(void (*k)(double*, double), int grid, double* d_out, double flops_per_block, const char* name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_out, 1.0);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_out, 1.0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tf = flops_per_block * grid / (best * 1e-3) / 1e12;
    printf("%-28s grid %5d x 256 threads  %8.3f ms  %7.2f TFLOP/s\n", name, grid, best, tf);
    return tf;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, grid = cus * 8;          // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    double* d_out; hipMalloc(&d_out, sizeof(double) * grid * 256);
    printf("device: %s, %d CUs, %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1e3);
    const double valu = run(k_valu, grid, d_out, 256.0 * 8 * 2 * ITERS, "v_fma_f64 (vector pipe)");
    const double mfma = run(k_mfma, grid, d_out, 4.0 * 4 * 2048 * ITERS, "v_mfma_f64_16x16x4_f64");
    printf("ratio matrix / vector fp64 rate: %.2f\n", mfma / valu);
    printf("useful fraction of 16x16 tiles on the 40x40 lower triangle of the condensed Hessian (820 of 9 tiles x 256 = 2304 entries of the\n"
           "tiles that touch it; the trailing update of an LDL^T sweeps shrinking sub-triangles): <= %.0f %%\n", 100.0 * 820 / 2304);
    printf("=> effective matrix-pipe rate on this factorisation <= %.1f TFLOP/s against %.1f TFLOP/s on the vector pipe with exact triangles\n",
           mfma * 820 / 2304, valu);
    hipFree(d_out);
    return 0;
}
