// Accuracy of v_rcp_f64 / v_rsq_f64 and of 1 / 2 Newton steps on top (relative error vs correctly rounded 1/x), gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d); r0[i] = r;
    double e = fma(-d, r, 1.0); r = fma(r, e, r); r1[i] = r;
    e = fma(-d, r, 1.0); r = fma(r, e, r); r2[i] = r;
}
int main() {
    const int n = 1 << 20; double *x, *a, *b, *c; hipMalloc(&x, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
    double* h = (double*)malloc(n * 8); srand(7);
    for (int i = 0; i < n; ++i) { double m = 1.0 + rand() / (double)RAND_MAX; int ex = rand() % 200 - 100; h[i] = ldexp(m, ex); }
    hipMemcpy(x, h, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(x, a, b, c, n);
    double *ha = (double*)malloc(n * 8), *hb = (double*)malloc(n * 8), *hc = (double*)malloc(n * 8);
    hipMemcpy(ha, a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, c, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)h[i];
        e0 = fmax(e0, (double)fabsl(((long double)ha[i] - t) / t)); e1 = fmax(e1, (double)fabsl(((long double)hb[i] - t) / t)); e2 = fmax(e2, (double)fabsl(((long double)hc[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err %.3e ; + 1 Newton step %.3e ; + 2 Newton steps %.3e  (2^-53 = %.3e)\n", e0, e1, e2, ldexp(1.0, -53));
    return 0;
}
