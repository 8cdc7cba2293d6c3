// A/B on the one GEMM-shaped piece of the car path (BASELINE.json north_star: "MFMA used only for the dense Nx x Nu stage-Hessian
// contractions"): the condensed Hessian H = sum_k Gamma_k' Q Gamma_k of the N = 20 step -- a 40 x 60 x 40
// SYRK per instance (three weighted state components x 20 stages) -- accumulated
//   V  as phase C of admpc_fused20.hip does it: lane i owns row i of H; per stage the three weighted components of Gamma_k are
//      published in LDS, read back as 16-lane blocks and folded in by v_fmac_f64_dpp row_newbcast (only the 2k inputs of earlier stages
//      are non-zero: 1260 FMAC instructions per instance), then the packed lower-triangular row store;
//   M  on the matrix pipe: six 16 x 16 tiles (I >= J) of v_mfma_f64_16x16x4_f64, K = 4 per stage (three components + a zero row),
//      tiles whose columns are still structurally zero skipped (91 MFMA instructions per instance), operands read from the same LDS
//      exchange buffer, result scattered from the MFMA layout into the same packed rows.
// Both kernels run the same synthetic Gamma recurrence (a few FMAs per stage standing in for the propagation), one instance per wave,
// persistent waves, and must produce the same H (checked on the host).  Run under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES for the
// matrix-pipe evidence:   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ad_mpc_amd/csrc -o mfma_condense_probe mfma_condense_probe.hip
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define WAVE 64
namespace {
#include "dense40.h"
}
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int N = 20, n = 40, NTRI = n * (n + 1) / 2;

// synthetic Gamma: component c of column `lane` at stage k (zero until the column's own stage, as in the real recursion)
__device__ __forceinline__ void gamma_step(double (&g)[3], int k, int lane, double seed) {
    const bool mine = (lane >> 1) == k;              // the two inputs of stage k pick up "B_k"
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double b = 0.01 * (c + 1) + 1e-3 * lane + seed;
        g[c] = mine ? b : 0.97 * g[c] + 0.02 * g[(c + 1) % 3];
    }
}

__global__ __launch_bounds__(WAVE, 2) void k_valu(int B, double* __restrict__ Hg, double seed)
{
    __shared__ double Hp[NTRI + 4], gam[3 * 64];
    const int lane = threadIdx.x;
    const bool uact = lane < n;
    const double w[3] = { 0.5, 0.5, 5.0 };
    for (int inst = blockIdx.x; inst < B; inst += gridDim.x) {
        double g[3] = { 0.0, 0.0, 0.0 }, hrow[n];
#pragma unroll
        for (int i = 0; i < n; ++i) hrow[i] = 0.0;
        static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            constexpr int lim = 2 * k < n ? 2 * k : n;
            constexpr int nblk = (lim + 15) / 16;
            int tok = B; asm volatile("" : "+s"(tok));
            if (tok > 0) {
                double wg[3], Rb[3][3];
                if constexpr (k >= 1) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { wg[c] = w[c] * g[c]; gam[c * 64 + lane] = g[c]; }
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int m = 0; m < nblk; ++m) Rb[c][m] = gam[c * 64 + 16 * m + (lane & 15)];
                }
                if constexpr (k < N) gamma_step(g, k, lane, seed + 1e-6 * inst);
                if constexpr (k >= 1) {
                    static_for<0, 3>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        static_for<0, lim / 4>([&](auto q) __attribute__((always_inline)) {
                            constexpr int i2 = 4 * decltype(q)::value;
                            fmac_rowbc4_ld<i2 % 16>(hrow[i2], hrow[i2 + 1], hrow[i2 + 2], hrow[i2 + 3], Rb[c][i2 / 16], wg[c]);
                        });
                        if constexpr (lim % 4 == 2) {
                            fmac_rowbc_ld<(lim - 2) % 16>(hrow[lim - 2], Rb[c][(lim - 2) / 16], wg[c]);
                            fmac_rowbc_ld<(lim - 1) % 16>(hrow[lim - 1], Rb[c][(lim - 1) / 16], wg[c]);
                        }
                    });
                }
            }
        });
        store_row_40(hrow, lds_byte_addr(Hp + (uact ? (lane * (lane + 1)) / 2 : 0)));
        WSYNC();
        for (int i = lane; i < NTRI; i += WAVE) Hg[(size_t)inst * NTRI + i] = Hp[i];
        WSYNC();
    }
}

__global__ __launch_bounds__(WAVE, 2) void k_mfma(int B, double* __restrict__ Hg, double seed)
{
    __shared__ double Hp[NTRI + 4], gam[4 * 64];
    const int lane = threadIdx.x;
    const int r16 = lane & 15, kq = lane >> 4;           // MFMA operand layout: lane = index + 16 * k, k = 0..3 (component; 3 = zero row)
    const double w[3] = { 0.5, 0.5, 5.0 };
    const double wk = kq < 3 ? w[kq < 3 ? kq : 0] : 0.0;
    gam[3 * 64 + lane] = 0.0;                            // the zero row of every K = 4 step
    WSYNC();
    for (int inst = blockIdx.x; inst < B; inst += gridDim.x) {
        double g[3] = { 0.0, 0.0, 0.0 };
        d4 acc[3][3];
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J < 3; ++J) acc[I][J] = d4{0.0, 0.0, 0.0, 0.0};
        static_for<0, N + 1>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            constexpr int lim = 2 * k < n ? 2 * k : n;
            constexpr int nblk = (lim + 15) / 16;            // column blocks of Gamma that are non-zero at this stage
            int tok = B; asm volatile("" : "+s"(tok));
            if (tok > 0) {
                double blk[3];
                if constexpr (k >= 1) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) gam[c * 64 + lane] = g[c];
                    // operand of block m: element (component kq, column 16 m + r16)
#pragma unroll
                    for (int m = 0; m < nblk; ++m) blk[m] = gam[kq * 64 + 16 * m + r16];
                }
                if constexpr (k < N) gamma_step(g, k, lane, seed + 1e-6 * inst);
                if constexpr (k >= 1) {
                    // H[I][J] += (W G_I)' G_J for the tiles whose rows AND columns are non-zero already: I, J < nblk, I >= J
                    static_for<0, nblk>([&](auto Ic) __attribute__((always_inline)) {
                        constexpr int I = decltype(Ic)::value;
                        const double a = wk * blk[I];
                        static_for<0, I + 1>([&](auto Jc) __attribute__((always_inline)) {
                            constexpr int J = decltype(Jc)::value;
                            acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, blk[J], acc[I][J], 0, 0, 0);
                        });
                    });
                }
            }
        });
        // scatter the tiles into the packed lower-triangular rows: D element (row = kq + 4 v, col = r16) of tile (I, J)
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * I + kq + 4 * v, col = 16 * J + r16;
                    if (row < n && col <= row) Hp[row * (row + 1) / 2 + col] = acc[I][J][v];
                }
        WSYNC();
        for (int i = lane; i < NTRI; i += WAVE) Hg[(size_t)inst * NTRI + i] = Hp[i];
        WSYNC();
    }
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, reps = 20;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * 8 < B ? prop.multiProcessorCount * 8 : B;
    double *hv, *hm;
    hipMalloc(&hv, (size_t)B * NTRI * 8); hipMalloc(&hm, (size_t)B * NTRI * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float tv = 1e30f, tm = 1e30f;
    for (int r = 0; r < reps; ++r) {
        float ms;
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k_valu, dim3(grid), dim3(WAVE), 0, 0, B, hv, 0.25); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (r > 1 && ms < tv) tv = ms;
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(WAVE), 0, 0, B, hm, 0.25); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (r > 1 && ms < tm) tm = ms;
    }
    std::vector<double> a((size_t)B * NTRI), b((size_t)B * NTRI);
    hipMemcpy(a.data(), hv, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), hm, b.size() * 8, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs(a[i] - b[i])); mx = fmax(mx, fabs(a[i])); }
    printf("condensed-Hessian SYRK (40 x 60 x 40 per instance), B = %d instances, grid %d one-wave workgroups (two waves per SIMD)\n", B, grid);
    printf("  V  v_fmac_f64_dpp rows (phase C of admpc_fused20.hip): %8.2f us per launch  (%.2f us per instance-wave)\n", tv * 1e3, tv * 1e3 * grid / B);
    printf("  M  v_mfma_f64_16x16x4_f64 tiles                      : %8.2f us per launch  (%.2f us per instance-wave)\n", tm * 1e3, tm * 1e3 * grid / B);
    printf("  ratio M / V = %.2f ; max |H_V - H_M| = %.2e (max |H| %.2e)\n", tm / tv, md, mx);
    return md <= 1e-12 * (1 + mx) ? 0 : 1;
}
