// argmin_rule.h -- the ONE statement of the scenario arg-min's ordering rules (BASELINE.json north_star: "RCCL ... only for the
// final arg-min reduction").  Compiled for the device (admpc_argmin_kernel, admpc_argmin_pairs_kernel) and for the host
// (admpc_argmin_pairs_host, the second-level reducer of hosts whose gathered records live in host memory, e.g. a gloo / MPI
// all-gather), so that both sides cannot drift apart.  tests/argmin_spec.py holds the table both are checked against.
//
//   * a NaN cost never wins: it is read as +inf;
//   * lower cost wins; equal costs -> the lower (global) index wins;
//   * nothing finite and no index at all (empty input, or every record NaN / +inf with index INT64_MAX) -> (+inf, index 0).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define ADMPC_HD __host__ __device__ __forceinline__
#else
#define ADMPC_HD static inline
#endif

struct ArgminBest { double v; int64_t i; };

ADMPC_HD ArgminBest argmin_identity() { ArgminBest b; b.v = INFINITY; b.i = INT64_MAX; return b; }
ADMPC_HD double argmin_cost(double c) { return c == c ? c : (double)INFINITY; }
// fold the candidate (cost already passed through argmin_cost, its index) into the running best
ADMPC_HD void argmin_fold(ArgminBest& b, double c, int64_t i) { if (c < b.v || (c == b.v && i < b.i)) { b.v = c; b.i = i; } }
ADMPC_HD int64_t argmin_final_index(const ArgminBest& b) { return b.i == INT64_MAX ? (int64_t)0 : b.i; }
