// Does a chain of dependent v_fmac_f64_dpp (DPP source = the accumulator itself, row_newbcast) need wait states on gfx950?
// In-row forward substitution on 16 lanes: y_i -= L[i][j] * y_j for i > j, j = 0..14, once with and once without s_nop pads,
// against a v_readlane reference.  Each 16-lane row of the wave solves its own system.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ double rdlane(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int J, bool PAD> __device__ __forceinline__ void step(double& y, const double nl) {
    // exec: lanes J..15 of every row (16-bit pattern replicated); lane J multiplies by 0
    constexpr unsigned long long row = (0xFFFFull << J) & 0xFFFFull;      // lanes J..15: the DPP source lane J must be enabled (no fetch-inactive on gfx9-class DPP)
    constexpr unsigned long long m = row | (row << 16) | (row << 32) | (row << 48);
    if (PAD) asm volatile("s_mov_b64 exec, %2\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1" : "+v"(y) : "v"(nl), "s"(m), "n"(J));
    else     asm volatile("s_mov_b64 exec, %2\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1" : "+v"(y) : "v"(nl), "s"(m), "n"(J));
}
template <bool PAD> __device__ double solve(double y, const double* Lrow) {   // Lrow[j] = L[lane%16][j]
#define S(J) step<J, PAD>(y, ((int)(threadIdx.x & 15) == J) ? 0.0 : -Lrow[J]);
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14)
#undef S
    return y;
}
__global__ void k(const double* Lg, const double* yg, double* o_pad, double* o_nopad, double* o_ref) {
    const int lane = threadIdx.x, r = lane & 15, base = lane & ~15;
    double Lrow[16];
    for (int j = 0; j < 16; ++j) Lrow[j] = Lg[(size_t)(blockIdx.x * 64 + lane) * 16 + j];
    const double y0 = yg[blockIdx.x * 64 + lane];
    o_pad[blockIdx.x * 64 + lane] = solve<true>(y0, Lrow);
    o_nopad[blockIdx.x * 64 + lane] = solve<false>(y0, Lrow);
    double y = y0;
    for (int j = 0; j < 15; ++j) {
        double zj[4]; for (int q = 0; q < 4; ++q) zj[q] = rdlane(y, 16 * q + j);
        const double z = zj[lane >> 4];
        if (r > j) y = fma(-Lrow[j], z, y);
    }
    o_ref[blockIdx.x * 64 + lane] = y; (void)base;
}
int main() {
    const int B = 4096; double *L, *y, *a, *b, *c; hipMalloc(&L, B * 64 * 16 * 8); hipMalloc(&y, B * 64 * 8); hipMalloc(&a, B * 64 * 8); hipMalloc(&b, B * 64 * 8); hipMalloc(&c, B * 64 * 8);
    double* h = (double*)malloc(B * 64 * 16 * 8); srand(3);
    for (int i = 0; i < B * 64 * 16; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * 0.4;
    hipMemcpy(L, h, B * 64 * 16 * 8, hipMemcpyHostToDevice);
    for (int i = 0; i < B * 64; ++i) h[i] = rand() / (double)RAND_MAX;
    hipMemcpy(y, h, B * 64 * 8, hipMemcpyHostToDevice);
    long bp = 0, bn = 0;
    double *ha = (double*)malloc(B * 64 * 8), *hb = (double*)malloc(B * 64 * 8), *hc = (double*)malloc(B * 64 * 8);
    for (int it = 0; it < 10; ++it) {
        k<<<B, 64>>>(L, y, a, b, c);
        hipMemcpy(ha, a, B * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, b, B * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, c, B * 64 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < B * 64; ++i) { if (ha[i] != hc[i]) ++bp; if (hb[i] != hc[i]) ++bn; }
    }
    printf("dpp chain probe: padded variant differs in %ld entries, unpadded in %ld (of %d)\n", bp, bn, 10 * B * 64);
    return 0;
}
