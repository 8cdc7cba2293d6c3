// admpc_rowqp.hip -- kernel R: the QP of one RTI step (H2-H6) for every horizon, fp64 and fp32.
//
// One MPC instance per 16-lane DPP row, four instances per wavefront (workgroup = one wave).  The algorithm is
// rowqp_core.h (shared with the CPU lane emulator of tests/emu), the lane primitives are rowqp_dev.h.
//
// Data placement
//   HBM / L2, read-only during the solve (streamed one stage ahead by every sweep): packed linearisation GT [B][N][7][6] and
//     defects b [B][N][7] from the linearisation kernel, the references yref / yref_e; xbar / ubar at the start and the end.
//   workspace [B][N+1][38] in global memory (L2 / Infinity Cache resident): what only the sweeps touch -- the absolute state,
//     its Newton step, the feedback gains, the corrector's sigma-mu coefficients -- written by one sweep, streamed by the next.
//   LDS: what the passes over the inequalities touch, 31 values per stage and instance (slacks, multipliers, inputs and their
//     references, right-hand sides and steps of the inputs / the steering angle): 248 B (fp64) / 124 B (fp32) per stage ->
//     N = 40 fp64 / N = 80 fp32: 9.9 KB per instance, four waves (16 instances) per CU, one wave per SIMD.
//   Registers: the Riccati matrix P (7 values per lane), the stage data of the current and the next stage, lane constants.
// Waves draw quadruples of instances from a ticket counter (zeroed by the linearisation kernel).  The four instances of a wave
// iterate until the last of them has converged; finished rows are frozen by masks (their state is not rewritten).
//
// Split batches (more quadruples than wave slots): launch 1 (mode 1) runs roll-out + unconstrained trial for every instance and
// finishes those the trial solves (42-55 % of the bench scenarios; every row of a wave does the same work); the others are
// deferred with a key = number of bounds their trial minimiser violates.  admpc_rowqp_sort_kernel orders them by key, hardest
// first.  Launch 2 (mode 2) draws quadruples from that list, restores their LDS regions (saved by launch 1) and continues behind the
// trial, exactly where the one-launch solve would be: rows of one wave now need similar
// iteration counts and the slow instances start first (a wave iterates until its slowest row has converged -- in ticket order
// the mean of the per-wave maxima is 8.5 iterations against a mean of 4.5 per instance).
#include "../../include/admpc.h"
#include "rowqp_dev.h"

namespace {

template <class T>
__global__ __launch_bounds__(64) void admpc_rowqp_kernel(const AdmpcConfig* __restrict__ cfg, int B, int rows, int inst_stride,
                                                         const T* __restrict__ x0g, const T* __restrict__ yrefg, const T* __restrict__ yrefeg,
                                                         const T* __restrict__ GTg, const T* __restrict__ blg,
                                                         T* __restrict__ xbarg, T* __restrict__ ubarg,
                                                         T* __restrict__ costg, int32_t* __restrict__ statusg, int32_t* __restrict__ itersg,
                                                         T* __restrict__ pig, T* __restrict__ ineqg, T* __restrict__ wsg, int first_pass, int* __restrict__ ticket,
                                                         int mode, int32_t* __restrict__ keyg, const int32_t* __restrict__ permg, const int32_t* __restrict__ countg,
                                                         T* __restrict__ dumpg)
{
    typedef DevX<T> X;
    extern __shared__ double smem_raw[];
    typename X::LT* const smem = (typename X::LT*)smem_raw;
    const int lane16 = (int)(threadIdx.x & 15u), row = (int)(threadIdx.x >> 4);
    RqParams<T> q;
    rq_make_params<T>(*cfg, q);
    RqArrays<T> io;
    io.x0 = x0g; io.yref = yrefg; io.yref_e = yrefeg; io.GT = GTg; io.bl = blg; io.xbar = xbarg; io.ubar = ubarg; io.pi = pig; io.ineq = ineqg; io.ws = wsg; io.dump = dumpg;
    const int total = mode == 2 ? __builtin_amdgcn_readfirstlane(*countg) : B;       // mode 2: the deferred instances, permg[0 .. total)
    const int nquads = (total + rows - 1) / rows;
    const bool has_lds = row < rows;
    typename X::Lds lds{ smem + (has_lds ? row : 0) * inst_stride, has_lds };

    for (int quad = (int)blockIdx.x; quad < nquads;) {
        const int inst = quad * rows + row;
        const bool owns = has_lds && inst < total;
        bool valid = owns;
        int ic = inst < total && has_lds ? inst : total - 1;                // rows without an instance recompute the last one and write nothing
        if (mode == 2) ic = permg[ic];
        if (!first_pass) valid = valid && statusg[ic] == 0;                 // failed in an earlier SQP iteration: left untouched
        RowQp<X> S(q, io, lds, ic, owns);
        typename RowQp<X>::Result res;
        S.solve(valid, res, pig != nullptr, valid, mode);
        if (mode == 1 && owns && lane16 == 0) {                             // 0: done (or not to be solved); 1..255: deferred
            int kv = (int)res.nviol; kv = kv < 1 ? 1 : (kv > 255 ? 255 : kv);
            keyg[ic] = res.deferred ? kv : 0;
        }
        valid = valid && !res.deferred;
        bool failed = res.failed;
        T J;
        S.finish(valid, failed, J);
        if (valid && lane16 == 0) {
            statusg[ic] = failed ? ADMPC_STATUS_QP_FAILURE : ADMPC_STATUS_SUCCESS;
            if (costg) costg[ic] = failed ? (T)INFINITY : J;
            if (itersg) itersg[ic] = res.iters;
        }
        X::fence();
        int v = 0;
        if (threadIdx.x == 0) v = atomicAdd(ticket, 1);
        quad = (int)gridDim.x + __builtin_amdgcn_readfirstlane(v);
    }
}

// Orders the deferred instances of a split batch by key, largest first (counting sort, one workgroup; the order inside a bucket is
// whatever the atomics give -- it only decides who shares a wave, results do not depend on it), and re-arms the ticket counter.
__global__ __launch_bounds__(1024) void admpc_rowqp_sort_kernel(int B, const int32_t* __restrict__ key, int32_t* __restrict__ perm,
                                                                int32_t* __restrict__ count, int* __restrict__ ticket)
{
    __shared__ int hist[256], base[256];
    const int tid = (int)threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < B; i += 1024) { const int k = key[i]; if (k > 0) atomicAdd(&hist[k], 1); }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int b = 255; b >= 1; --b) { base[b] = acc; acc += hist[b]; }
        *count = acc; *ticket = 0;
    }
    __syncthreads();
    for (int i = tid; i < B; i += 1024) { const int k = key[i]; if (k > 0) perm[atomicAdd(&base[k], 1)] = i; }
}

template <class T>
static void rowqp_launch(int grid, int lds_bytes, hipStream_t st, const AdmpcConfig* d_cfg, int B, int rows, int inst_stride,
                         const T* x0, const T* yref, const T* yref_e, const T* GT, const T* bl, T* xbar, T* ubar, T* cost, int32_t* stat, int32_t* iters,
                         T* pi, T* ineq, T* ws, int first, int* ticket, int32_t* split, T* dump)
{
    if (!split) {
        hipLaunchKernelGGL((admpc_rowqp_kernel<T>), dim3(grid), dim3(64), lds_bytes, st, d_cfg, B, rows, inst_stride, x0, yref, yref_e, GT, bl,
                           xbar, ubar, cost, stat, iters, pi, ineq, ws, first, ticket, 0, (int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (T*)nullptr);
        return;
    }
    int32_t* key = split; int32_t* perm = split + B; int32_t* count = split + 2 * (size_t)B;
    hipLaunchKernelGGL((admpc_rowqp_kernel<T>), dim3(grid), dim3(64), lds_bytes, st, d_cfg, B, rows, inst_stride, x0, yref, yref_e, GT, bl,
                       xbar, ubar, cost, stat, iters, pi, ineq, ws, first, ticket, 1, key, (const int32_t*)nullptr, (const int32_t*)nullptr, dump);
    hipLaunchKernelGGL(admpc_rowqp_sort_kernel, dim3(1), dim3(1024), 0, st, B, (const int32_t*)key, perm, count, ticket);
    hipLaunchKernelGGL((admpc_rowqp_kernel<T>), dim3(grid), dim3(64), lds_bytes, st, d_cfg, B, rows, inst_stride, x0, yref, yref_e, GT, bl,
                       xbar, ubar, cost, stat, iters, pi, ineq, ws, first, ticket, 2, (int32_t*)nullptr, (const int32_t*)perm, (const int32_t*)count, dump);
}

}  // namespace

// LDS bytes of one instance: (N + 1) records, padded so that consecutive rows of a wave fall on different bank halves
// (ds_read_b64: lanes 0-31 = rows 0, 1 share one access; 16 consecutive doubles = 32 banks).
template <class T>
static int rowqp_inst_stride(int N)
{
    int s = RQ_HDR + N * RQ_RS;                                             // values of T
    s += s & 1;
    const int half = sizeof(T) == 8 ? 16 : 16;                              // fp64: 16 doubles = 128 B; fp32: 16 floats = 64 B
    while (s % (2 * half) != half) s += 2;
    return s;
}

extern "C" __attribute__((visibility("hidden")))
int admpc_rowqp_plan(int N, int elem, int B, int num_cu, int* rows, int* inst_stride, int* lds_bytes, int* grid)
{
    const int stride = elem == 8 ? rowqp_inst_stride<double>(N) : rowqp_inst_stride<float>(N);
    const int cap = 160 * 1024;
    int r = 4;
    while (r > 1 && r * stride * elem > cap) r >>= 1;
    if (r * stride * elem > cap) return -1;
    // small batches: fewer instances per wave spread the work over more SIMDs (one wave per SIMD before rows are shared)
    while (r > 1 && (B + r - 1) / r < num_cu * 4 && (B + r / 2 - 1) / (r / 2) <= num_cu * 4) r >>= 1;
    int per_cu = cap / (r * stride * elem);
    if (per_cu > 4) per_cu = 4;          // one wave per SIMD whatever the LDS allows: 256 VGPR + 148 (fp64) / 64 (fp32) AGPR per wave.  The grid must be the
                                         // number of waves that RUN at once: the caller counts rounds with it (split batches)
    int g = num_cu * per_cu;
    const int nquads = (B + r - 1) / r;
    if (g > nquads) g = nquads;
    *rows = r; *inst_stride = stride; *lds_bytes = r * stride * elem; *grid = g < 1 ? 1 : g;
    return 0;
}

extern "C" __attribute__((visibility("hidden")))
void admpc_rowqp_prepare(void)
{
    (void)hipFuncSetAttribute((const void*)admpc_rowqp_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)admpc_rowqp_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// split: nullptr = one launch; otherwise [2 B + 1] ints of scratch (keys, order, count) and the batch goes through the two phases above;
// dump: [B][16 + 31 N] values, the LDS regions of the deferred instances between the phases
extern "C" __attribute__((visibility("hidden")))
void admpc_rowqp_launch_f64(int grid, int lds_bytes, hipStream_t st, const AdmpcConfig* d_cfg, int B, int rows, int inst_stride,
                            const double* x0, const double* yref, const double* yref_e, const double* GT, const double* bl,
                            double* xbar, double* ubar, double* cost, int32_t* stat, int32_t* iters, double* pi, double* ineq, double* ws, int first, int* ticket,
                            int32_t* split, double* dump)
{
    rowqp_launch<double>(grid, lds_bytes, st, d_cfg, B, rows, inst_stride, x0, yref, yref_e, GT, bl, xbar, ubar, cost, stat, iters, pi, ineq, ws, first, ticket, split, dump);
}

extern "C" __attribute__((visibility("hidden")))
void admpc_rowqp_launch_f32(int grid, int lds_bytes, hipStream_t st, const AdmpcConfig* d_cfg, int B, int rows, int inst_stride,
                            const float* x0, const float* yref, const float* yref_e, const float* GT, const float* bl,
                            float* xbar, float* ubar, float* cost, int32_t* stat, int32_t* iters, float* pi, float* ineq, float* ws, int first, int* ticket,
                            int32_t* split, float* dump)
{
    rowqp_launch<float>(grid, lds_bytes, st, d_cfg, B, rows, inst_stride, x0, yref, yref_e, GT, bl, xbar, ubar, cost, stat, iters, pi, ineq, ws, first, ticket, split, dump);
}

#ifdef ADMPC_PHASE_TIMERS
#include <cstdio>
extern "C" void admpc_rowqp_dump_timers(void)
{
    unsigned long long h[16] = {0};
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rq_ticks), sizeof h) != hipSuccess) return;
    static const char* nm[16] = {"-", "S0 rollout", "trial+init", "E1", "check", "S1 factor", "S2 forward", "E2", "S3 backward", "S4 forward", "E3a", "E3b+E1",
                                 "-", "SF rollout", "-", "-"};
    unsigned long long tot = 0; for (int i = 0; i < 14; ++i) tot += h[i];
    for (int i = 1; i < 14; ++i) if (h[i]) fprintf(stderr, "[rowqp phase] %-12s %14llu ticks %5.1f %%\n", nm[i], h[i], 100.0 * (double)h[i] / (double)(tot ? tot : 1));
    if (h[15]) fprintf(stderr, "[rowqp phase] of which inside the %llu phase fences %llu ticks %5.1f %% (%.0f ticks per fence)\n", h[15], h[14],
                       100.0 * (double)h[14] / (double)(tot ? tot : 1), (double)h[14] / (double)h[15]);
}
#endif
