// Probe of gfx950 cross-lane primitives used by the DPP Cholesky: v_permlane16_swap, v_permlane32_swap, v_fmac_f64_dpp row_newbcast.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o, double* od) {
    const unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, x + 100, false, false);     // (vdst_old, src0_old)
    auto q = __builtin_amdgcn_permlane32_swap(x, x + 100, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
    double acc = 1000.0, l = (double)threadIdx.x, s = 2.0;
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(l), "v"(s));
    od[threadIdx.x] = acc;
}
int main() {
    unsigned* o; double* od; hipMalloc(&o, 256 * 4); hipMalloc(&od, 64 * 8);
    k<<<1, 64>>>(o, od); unsigned h[256]; double hd[64];
    hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hd, od, sizeof hd, hipMemcpyDeviceToHost);
    const char* nm[4] = {"p16 ret0", "p16 ret1", "p32 ret0", "p32 ret1"};
    for (int a = 0; a < 4; ++a) { printf("%s:", nm[a]); for (int i = 0; i < 64; ++i) printf(" %u", h[a * 64 + i]); printf("\n"); }
    printf("fmac dpp:"); for (int i = 0; i < 64; ++i) printf(" %g", hd[i]); printf("\n");
    return 0;
}
