// Probe: what one vector-memory load instruction costs a LONE wave on gfx950 for the access shapes of kernel R
// (four instances per wave, one per 16-lane row, each row reading a few consecutive doubles of its own instance).
//   hipcc --offload-arch=gfx950 -O3 -o vmem_probe vmem_probe.hip && ./vmem_probe
// One workgroup of one wave per CU (256 blocks), every wave streams over its own L2-resident region; reports ticks per load
// instruction (s_memtime around a stream of 2048 independent loads: up to 64 in flight, waited for once at the end).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>

__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// MODE 0: dwordx2, lanes 0..5 of each row consecutive doubles, other lanes duplicate lane 0 (what load_gr does)
// MODE 1: same addresses, lanes >= 6 of each row switched off (EXEC)
// MODE 2: dwordx2, only lanes 7, 8 distinct (K rows), others duplicates
// MODE 3: dwordx2, only lanes 7, 8 active
// MODE 4: dwordx4 per lane: lane c reads 16 B at c*48 (column layout, first of three)
// MODE 5: dwordx2 fully coalesced (64 consecutive doubles)
// MODE 6: dwordx4 fully coalesced
// MODE 7: dwordx2, 14 lanes of each row consecutive doubles (K0 | K1 in one instruction), others off
// MODE 8: dwordx2, all 16 lanes of each row consecutive doubles
template <int MODE>
__global__ __launch_bounds__(64) void k(const double* __restrict__ buf, size_t wave_stride, size_t row_stride, double* out, unsigned long long* cyc, int reps, int wrap)
{
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    const char* base = (const char*)(buf + blockIdx.x * wave_stride + row * row_stride);
    int off;
    bool act = true;
    if (MODE == 0 || MODE == 1) { off = (lane < 6 ? lane : 0) * 8; act = MODE == 0 || lane < 6; }
    else if (MODE == 2 || MODE == 3) { off = (lane == 8 ? 7 : 0) * 8; act = MODE == 2 || lane == 7 || lane == 8; }
    else if (MODE == 4) off = (lane < 9 ? lane : 0) * 48;
    else if (MODE == 5) off = (int)threadIdx.x * 8 - row * (int)(row_stride * 8);
    else if (MODE == 6) off = (int)threadIdx.x * 16 - row * (int)(row_stride * 8);
    else if (MODE == 7) { off = (lane < 14 ? lane : 0) * 8; act = lane < 14; }
    else off = lane * 8;
    double acc = 0;
    // consecutive loads of a row advance by 48 B (the next column / the next part of a record: cache lines are re-used as in the
    // kernel); the coalesced shapes advance by the bytes one instruction covers
    constexpr int STEP = MODE == 5 ? 512 : (MODE == 6 ? 1024 : 48);
    // compiler-visible loads, software-pipelined: batch r + 1 (32 loads) is issued before batch r is summed -> 32..64 loads in flight
    typedef typename std::conditional<MODE == 4 || MODE == 6, double2, double>::type L;
    auto sum = [](L v) { if constexpr (MODE == 4 || MODE == 6) return v.x + v.y; else return v; };
    L va[32], vb[32];
    const unsigned long long t0 = now();
    if (act) {
        const char* p = base + off;
#pragma unroll
        for (int i = 0; i < 32; ++i) va[i] = *(const L*)(p + i * STEP);
        for (int r = 1; r < reps; r += 2) {
            p = base + (size_t)(r % wrap) * (32 * STEP) + off;
#pragma unroll
            for (int i = 0; i < 32; ++i) vb[i] = *(const L*)(p + i * STEP);
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += sum(va[i]);
            p = base + (size_t)((r + 1 < reps ? r + 1 : r) % wrap) * (32 * STEP) + off;
#pragma unroll
            for (int i = 0; i < 32; ++i) va[i] = *(const L*)(p + i * STEP);
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += sum(vb[i]);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += sum(va[i]);
    }
    const unsigned long long total = now() - t0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = total;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, const double* buf, double* out, unsigned long long* cyc, int waves_per_cu, int wrap)
{
    const int grid = 256 * waves_per_cu, reps = 64;
    const size_t wave_stride = 4 * 65536, row_stride = 65536;       // doubles: 512 KB per row region, 2 MB per wave
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, buf, wave_stride, row_stride, out, cyc, reps, wrap); (void)hipDeviceSynchronize(); }
    unsigned long long h[2048]; (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
    printf("%-74s %d wave(s)/CU, %s: %7.1f ticks per load instruction\n", name, waves_per_cu, wrap >= 64 ? "streaming (HBM)" : "re-reading 3 KB per row (L2)", s / grid / reps / 32);
}

int main()
{
    double* buf; double* out; unsigned long long* cyc;
    const size_t n = (size_t)1024 * 4 * 65536 + 65536;
    if (hipMalloc(&buf, n * 8) != hipSuccess || hipMemset(buf, 0, n * 8) != hipSuccess) { printf("allocation failed\n"); return 1; }
    if (hipMalloc(&out, 2048 * 64 * 8) != hipSuccess || hipMalloc(&cyc, 2048 * 8) != hipSuccess) { printf("allocation failed\n"); return 1; }
    for (int wrap = 64; wrap >= 2; wrap /= 32)
        for (int w = 1; w <= 4; w *= 4) {
            run<0>("x2: 6 consecutive doubles per row, other lanes duplicates", buf, out, cyc, w, wrap);
            run<1>("x2: 6 consecutive doubles per row, other lanes off", buf, out, cyc, w, wrap);
            run<3>("x2: lanes 7, 8 only", buf, out, cyc, w, wrap);
            run<4>("x4: 9 lanes per row, 16 B at a 48 B stride (column layout)", buf, out, cyc, w, wrap);
            run<5>("x2: 64 consecutive doubles", buf, out, cyc, w, wrap);
            run<6>("x4: 64 consecutive 16 B", buf, out, cyc, w, wrap);
            run<7>("x2: 14 consecutive doubles per row, others off", buf, out, cyc, w, wrap);
        }
    return 0;
}
