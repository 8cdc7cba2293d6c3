// Probe: where do the 2048 one-wave blocks of the interior-point kernel's launch shape (64 threads, 256 VGPRs, ~20 KB LDS) land?
// Prints, for the grid order the kernel uses, which block indices share a SIMD (HW_REG_HW_ID / HW_REG_XCC_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__(64, 2) void k(unsigned* out, unsigned long long* t0out, int spin) {
    extern __shared__ double lds[];
    unsigned hw, xcc; unsigned long long t0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)\n\ts_memrealtime %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(hw), "=s"(xcc), "=s"(t0));
    asm volatile("v_mov_b32 v255, 0" ::: "v255");        // claim 256 VGPRs like the real kernel
    double a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
    lds[threadIdx.x] = a;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; t0out[blockIdx.x] = t0; }
    if (a == 12345.678) out[0] = 0;
}
int main(int argc, char** argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 2048, lds = argc > 2 ? atoi(argv[2]) : 20480, spin = 20000;
    unsigned* d; unsigned long long* dt;
    hipMalloc(&d, grid * 8); hipMalloc(&dt, grid * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, 0, d, dt, spin); hipDeviceSynchronize(); }
    std::vector<unsigned> h(2 * grid); std::vector<unsigned long long> ht(grid);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> simd;
    unsigned long long tmin = *std::min_element(ht.begin(), ht.end());
    for (int b = 0; b < grid; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned simd_id = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned key = (xcc << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd_id;
        simd[key].push_back(b);
        if (b < 24 || (b >= 1024 && b < 1032)) printf("block %4d xcc %u se %u sh %u cu %2u simd %u wave %u t0 %llu\n", b, xcc, se, sh, cu, simd_id, hw & 0xf, ht[b] - tmin);
    }
    printf("distinct SIMDs used: %zu\n", simd.size());
    std::map<int, int> hist; std::map<int, int> gap;
    int shown = 0;
    for (auto& kv : simd) {
        hist[(int)kv.second.size()]++;
        if (kv.second.size() == 2) { int g = abs(kv.second[1] - kv.second[0]); gap[g]++; }
        if (shown < 12) { printf("simd key %05x blocks:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); ++shown; }
    }
    for (auto& kv : hist) printf("SIMDs with %d blocks: %d\n", kv.first, kv.second);
    printf("index gaps between the two blocks of a SIMD (gap: count), top:\n");
    std::vector<std::pair<int,int>> gv(gap.begin(), gap.end());
    std::sort(gv.begin(), gv.end(), [](auto& a, auto& b){ return a.second > b.second; });
    for (size_t i = 0; i < gv.size() && i < 12; ++i) printf("  %d: %d\n", gv[i].first, gv[i].second);
    return 0;
}
