// work_order.h -- the work order of the persistent condensed kernels (admpc_fused20.hip: one wave per instance; admpc_seg.hip: one
// workgroup of S waves per instance): a pre-pass bins the instances by a kinematic estimate of their interior-point effort, the
// persistent workgroups draw tickets that walk the bins from the most expensive down.  Include inside the translation unit's anonymous
// namespace after cond_common.h (wave_scan_incl_int) with NX defined.  A heuristic: it orders work and nothing else.
#pragma once

#define F20_NB 64
#define F20_BINS0 64
#define F20_HDR 128
#define F20_NQ 64           // sub-queues of the expansion queue (admpc_fused20.hip): a power of two, at most the wave size
// The estimate (round 4): the longitudinal input saturates when the speed of the reference -- the distance between its first two points
// per sampling interval -- differs from the vehicle's by more than the box allows the tracking controller to ask for: the acceleration an
// LQ tracker of these weights starts with is ~ k (v_ref - v_x), k = 3 / s here (a double integrator with the reference's weights 10 / 1
// has velocity gain 2.5; calibrated on the scenario generator: with k = 3 not one iterating instance of the N = 40 bench batch lands in
// the cheapest bin, with the earlier estimate 994 of 2767 did).  The earlier estimate -- the constant acceleration that reaches the
// along-track position of the TERMINAL reference, 2 (s - v_x T) / T^2 -- scales with 1 / T and projects a curved path onto the initial
// heading: calibrated at T = 1 s it under-estimates at T = 2 s and let expensive instances start last (F20_ORDER_TERMINAL=1 keeps it).
#ifndef F20_ORDER_TERMINAL
#define F20_ORDER_TERMINAL 0
#endif
#ifndef F20_ORDER_GAIN
#define F20_ORDER_GAIN 3.0
#endif
#ifdef F20_ORDER_HINT      // experiment builds only (scripts/experiments/order_headroom.py): bins from a per-instance effort the caller knows
__device__ const int* g_order_hint = nullptr;
#endif
__global__ __launch_bounds__(256) void admpc_f20_order_kernel(const AdmpcConfig* __restrict__ cfg, int B, const double* __restrict__ x0g,
                                                               const double* __restrict__ yrefg, const double* __restrict__ yrefeg,
                                                               int* __restrict__ sched, int cap, int* __restrict__ sched_next)
{
    // The handle keeps TWO scheduler states and alternates between them: this launch zeroes the header of the NEXT one (nobody is using it: the
    // previous launch of the stream is complete), so every launch pair starts from a clean header without a memset of its own in front of it
    // (2 us of a 200 us step at configs[1]), and a launch that failed half-way cannot poison the one after the next either.
    if (blockIdx.x == 0 && threadIdx.x < F20_HDR) sched_next[threadIdx.x] = 0;
    // Bin counts are aggregated per block in LDS and reach the global counters as ONE atomic per (block, bin): 2000 of 4096 config-2
    // instances share bin 0, and one global atomic each on that word took 22 us -- a tenth of the step.
    __shared__ int cnt[F20_NB], base[F20_NB];
    if (threadIdx.x < F20_NB) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int q = -1, rank = 0;
#if defined(F20_DEFER) && F20_DEFER
    // expansion queues of admpc_fused20_kernel (this launch's state): counters zero, entries not published
    for (int i = b; i < F20_NQ * 32 + cap + F20_NQ; i += (int)(gridDim.x * blockDim.x)) sched[F20_HDR + (size_t)F20_NB * cap + i] = i < F20_NQ * 32 ? 0 : -1;
#endif
    if (b < B) {
        const double* x0 = x0g + (size_t)b * NX;
#if F20_ORDER_TERMINAL
        const double* ye = yrefeg + (size_t)b * NX;
        const double T = cfg->Ts * (double)cfg->N;
        double sn, cs;
        sincos(x0[2], &sn, &cs);
        const double along = cs * (ye[0] - x0[0]) + sn * (ye[1] - x0[1]);
        const double areq = 2.0 * (along - x0[3] * T) / (T * T);
#else
        const double* yr = yrefg + (size_t)b * cfg->N * NY;
        const double ex = yr[NY] - yr[0], ey = yr[NY + 1] - yr[1];
        const double vref = sqrt(ex * ex + ey * ey) / cfg->Ts;
        const double areq = F20_ORDER_GAIN * (vref - x0[3]);
#endif
        const double lb = cfg->lbu[0], ub = cfg->ubu[0];
        const double ov = fmax(areq - ub, lb - areq) / (ub - lb);          // < 0: that far inside the box
        q = 0;
        if (ov == ov && ov > -0.125) q = 1 + (int)fmin(fmax((ov + 0.125) * 32.0, 0.0), (double)(F20_NB - 2));
#ifdef F20_ORDER_HINT
        if (g_order_hint) { const int hq = g_order_hint[b]; q = hq < 0 ? 0 : (hq > F20_NB - 1 ? F20_NB - 1 : hq); }
#endif
        rank = atomicAdd(&cnt[q], 1);
    }
    __syncthreads();
    if (threadIdx.x < F20_NB) base[threadIdx.x] = cnt[threadIdx.x] > 0 ? atomicAdd(sched + F20_BINS0 + threadIdx.x, cnt[threadIdx.x]) : 0;
    __syncthreads();
    // never past the bin's list: the counts start from zero in every launch (the launch functions clear the header on the caller's stream),
    // and a handle admits one solve at a time (admpc.h) -- should a caller break that rule, work is dropped from the ORDER, not memory overrun
    if (q >= 0 && base[q] + rank < cap) sched[F20_HDR + (size_t)q * cap + base[q] + rank] = b;
}

// next instance for a persistent wave (wave-uniform), -1 when the batch is drained: tickets walk the bins from the most expensive
// down.  First ticket = block index (2048 simultaneous atomics on one word queue up for ~20 us), later ones from the counter.
// cap == 0: no work order -- the batch fits the grid (one instance per workgroup, all start at once: nothing to order, and the launch
// function has not run the pre-pass: 5 - 8 us less latency for a small batch, the reference's own use is B = 1).
__device__ __forceinline__ int f20_next(int* __restrict__ sched, int cap, bool first, int lane) {
    if (cap == 0) return first ? (int)blockIdx.x : -1;
    int t = blockIdx.x;
#if defined(F20_PAIR_REV) && F20_PAIR_REV        // experiment: the second wave of a SIMD takes the EASIEST of the first round's tickets (block b + G/2 shares the SIMD of block b)
    if (t >= (int)gridDim.x / 2) t = (int)gridDim.x + (int)gridDim.x / 2 - 1 - t;
#endif
    if (!first) {
        int v = 0;
        if (lane == 0) v = atomicAdd(sched, 1);
        t = (int)gridDim.x + __builtin_amdgcn_readfirstlane(v);
    }
    const int c = sched[F20_BINS0 + F20_NB - 1 - lane];
    const int incl = wave_scan_incl_int(c);
    const unsigned long long m = __ballot(incl > t);
    if (m == 0ull) return -1;
    const int l = __ffsll((long long)m) - 1;
    const int base = __builtin_amdgcn_readlane(incl - c, l);
    return __builtin_amdgcn_readfirstlane(sched[F20_HDR + (size_t)(F20_NB - 1 - l) * cap + (t - base)]);
}

