import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config, tight_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
N, B = 80, 16384
cfg = default_config(N=N); eng = BatchSolver(cfg, device=0); o = Oracle(omp=True)
for seed in (1234, 100, 101, 102, 103, 104):
    for kw in ({}, {"blend": (3.0, 5.0)}):
        s = random_scenarios(B, N=N, seed=seed, **kw)
        g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], dtype=np.float32)
        r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=32)
        t = o.solve_batch(tight_config(N=N), s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=32)
        du_r = np.abs(g[1] - r[1]).max(axis=(1, 2)); du_t = np.abs(g[1] - t[1]).max(axis=(1, 2))
        print("seed %4d %s: vs default-level oracle max %.2e p99.9 %.2e | vs tight oracle max %.2e p99.9 %.2e p99 %.2e | dx max %.2e" % (seed, "dyn" if kw else "kin", du_r.max(), np.quantile(du_r, .999), du_t.max(), np.quantile(du_t, .999), np.quantile(du_t, .99), np.abs(g[0] - t[0]).max()), flush=True)
