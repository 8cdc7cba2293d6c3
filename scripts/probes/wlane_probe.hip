// Probe: is v_writelane_b32 -> v_readlane_b32 of the same VGPR (compiler-generated SGPR spill / reload pattern) safe back to back?
// Each wave repeats: fill v10/v11, write one lane with a tagged value, read back that lane / another lane at distance 0 and 1.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* bad, int iters) {
    unsigned errs0 = 0, errs1 = 0, errsA = 0, errsB = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned tag = 0x1000u + (unsigned)it * 7u + blockIdx.x;
        unsigned base = 0x5a5a0000u + (unsigned)it;
        unsigned r_same0, r_other0, r_same1, r_other1;
        asm volatile(
            "v_mov_b32 v10, %4\n\t"
            "v_mov_b32 v11, %4\n\t"
            "s_nop 7\n\t"
            "v_writelane_b32 v10, %5, 60\n\t"
            "v_readlane_b32 %0, v10, 60\n\t"
            "s_nop 7\n\t"
            "v_writelane_b32 v11, %5, 61\n\t"
            "v_readlane_b32 %1, v11, 29\n\t"
            "s_nop 7\n\t"
            "v_mov_b32 v10, %4\n\t"
            "v_mov_b32 v11, %4\n\t"
            "s_nop 7\n\t"
            "v_writelane_b32 v10, %5, 60\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %2, v10, 60\n\t"
            "s_nop 7\n\t"
            "v_writelane_b32 v11, %5, 61\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %3, v11, 29\n\t"
            "s_nop 7\n\t"
            : "=s"(r_same0), "=s"(r_other0), "=s"(r_same1), "=s"(r_other1) : "s"(base), "s"(tag) : "v10", "v11");
        errs0 += (r_same0 != tag); errsA += (r_other0 != base); errs1 += (r_same1 != tag); errsB += (r_other1 != base);
    }
    if (threadIdx.x == 0) { atomicAdd(bad + 0, errs0); atomicAdd(bad + 1, errsA); atomicAdd(bad + 2, errs1); atomicAdd(bad + 3, errsB); }
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 16); (void)hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, d, 2000); (void)hipDeviceSynchronize();
    unsigned h[4]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mismatches over 4096 waves x 2000 repetitions: same lane d0 %u, other lane d0 %u, same lane d1 %u, other lane d1 %u\n", h[0], h[1], h[2], h[3]);
    return 0;
}
