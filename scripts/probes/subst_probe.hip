// Isolation test of the generated substitution assembly (subst_asm.inc): many waves, each with its own packed unit-lower factor
// in LDS, run L z = y / L' x = z through the assembly and through plain loops; results must agree bit for bit, every repetition.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../ad_mpc_amd/csrc/subst_asm.inc"
#define N40 40
__device__ __forceinline__ unsigned lds_byte_addr(const double* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const double*)p; }
__device__ __forceinline__ double rdlane(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(64, 2) void k(const double* Lg, const double* yg, double* out_asm, double* out_ref, int reps) {
    extern __shared__ double Lp[];
    const int lane = threadIdx.x, trz = lane * (lane + 1) / 2;
    const bool uact = lane < N40;
    double* const pubb = Lp + 820;                                  // 64-double exchange buffer of the blocked substitution
    for (int i = lane; i < 820; i += 64) Lp[i] = Lg[(size_t)blockIdx.x * 820 + i];
    __syncthreads();
    if (uact) Lp[trz + lane] = 0.0;                                 // diagonal slots: the multiplier of a step's own source lane
    __syncthreads();
    for (int r = 0; r < reps; ++r) {
        double y = uact ? yg[(size_t)blockIdx.x * 64 + lane] + r : 0.0, yr = y;
        asm volatile(ADMPC_FWD_SUBST_ASM_40 : "+{v[100:101]}"(y) : "{v102}"(lds_byte_addr(Lp + (uact ? trz : 0))), "{v103}"(lds_byte_addr(pubb + (lane & 15))) : ADMPC_SUBST_CLOBBERS);
        double x = y;
        asm volatile(ADMPC_BWD_SUBST_ASM_40 : "+{v[100:101]}"(x) : "{v102}"(lds_byte_addr(Lp + (uact ? lane : 0))), "{v103}"(lds_byte_addr(pubb + (lane & 15))) : ADMPC_SUBST_CLOBBERS);
        for (int j = 0; j < N40 - 1; ++j) { const double zj = rdlane(yr, j); const double l = (uact && lane > j) ? Lp[trz + j] : 0.0; yr -= l * zj; }
        double xr = yr;
        for (int j = N40 - 1; j >= 1; --j) { const double xj = rdlane(xr, j); const double l = lane < j ? Lp[j * (j + 1) / 2 + lane] : 0.0; xr -= l * xj; }
        out_asm[((size_t)blockIdx.x * reps + r) * 64 + lane] = uact ? x : 0.0;
        out_ref[((size_t)blockIdx.x * reps + r) * 64 + lane] = uact ? xr : 0.0;
    }
}
int main() {
    const int B = 2048, reps = 8;
    double *L, *y, *oa, *orf; hipMalloc(&L, B * 820 * 8); hipMalloc(&y, B * 64 * 8); hipMalloc(&oa, (size_t)B * reps * 64 * 8); hipMalloc(&orf, (size_t)B * reps * 64 * 8);
    double* h = (double*)malloc(B * 820 * 8); srand(1);
    for (int i = 0; i < B * 820; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * 0.3;
    hipMemcpy(L, h, B * 820 * 8, hipMemcpyHostToDevice);
    for (int i = 0; i < B * 64; ++i) h[i] = rand() / (double)RAND_MAX;
    hipMemcpy(y, h, B * 64 * 8, hipMemcpyHostToDevice);
    double* a = (double*)malloc((size_t)B * reps * 64 * 8); double* b = (double*)malloc((size_t)B * reps * 64 * 8); double* a0 = (double*)malloc((size_t)B * reps * 64 * 8);
    long bad_ref = 0, bad_rep = 0;
    for (int it = 0; it < 20; ++it) {
        k<<<B, 64, (820 + 64) * 8>>>(L, y, oa, orf, reps);
        hipMemcpy(a, oa, (size_t)B * reps * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(b, orf, (size_t)B * reps * 64 * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < (size_t)B * reps * 64; ++i) { if (a[i] != b[i]) ++bad_ref; if (it > 0 && a[i] != a0[i]) ++bad_rep; }
        if (it == 0) for (size_t i = 0; i < (size_t)B * reps * 64; ++i) a0[i] = a[i];
    }
    printf("subst probe: %ld entries differ from the loop version, %ld differ between repetitions\n", bad_ref, bad_rep);
    return bad_ref || bad_rep;
}
