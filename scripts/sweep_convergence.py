"""Analysis script (uses the oracle -- not part of the product): iteration-count census of the car interior point over random
scenario batches, with and without cfg.ipm_fallback_iter.  Evidence for DESIGN.md section 3 (fallback mode).
    python3 scripts/sweep_convergence.py [seeds]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
o = Oracle(omp=True)
total = 0
for N, B in ((20, 4096), (40, 4096), (80, 2048), (128, 512)):
    for kw in ({}, {"blend": (3.0, 5.0)}):
        cfg = default_config(N=N); off = cfg.copy(); off.ipm_fallback_iter = 0.0
        mx = 0; healthy = 0; fb = []; stuck = []; bad = 0
        for seed in range(seeds):
            s = random_scenarios(B, N=N, seed=seed, **kw)
            r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
            r0 = o.solve_batch(off, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
            total += B; bad += int((r[3] != 0).sum())
            healthy = max(healthy, int(r0[4][r0[4] < cfg.ipm_iter_max].max()))
            mx = max(mx, int(r[4].max()))
            fb += [(seed, int(i), int(r[4][i])) for i in np.nonzero(r[4] > cfg.ipm_fallback_iter)[0]]
            stuck += [(seed, int(i)) for i in np.nonzero(r0[4] >= cfg.ipm_iter_max)[0]]
        print(f"N {N:3d} {'dynamic' if kw else 'kinematic':9s} {seeds * B:6d} scenarios: status!=0 {bad}; largest count of an instance that converges without the "
              f"fallback {healthy}; at iter_max without it {stuck}; entered the fallback (seed, index, iters) {fb}; max iters {mx}", flush=True)
print("total", total)
