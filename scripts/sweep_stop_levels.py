import sys, numpy as np
sys.path.insert(0, '/root/repo')
from ad_mpc_amd.config import default_config, tight_config
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
o = Oracle(omp=True)
for N, B in ((20, 8192), (40, 4096), (80, 4096), (128, 1024)):
    for seed in (7, 8):
        for kw in ({}, {"blend": (3.0, 5.0)}):
            s = random_scenarios(B, N=N, seed=seed, **kw)
            a = o.solve_batch(default_config(N=N), s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
            t = o.solve_batch(tight_config(N=N), s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
            ok = (a[3] == 0) & (t[3] == 0)
            dev = np.abs(a[1] - t[1]).max(axis=(1, 2))
            print("N %3d seed %d %s: status!=0 %d/%d  iters max %d/%d mean %.2f/%.2f  |du_def - du_tight| max %.1e p99.9 %.1e  more-iterations-than-tight %d" % (N, seed, "dyn" if kw else "kin", (a[3] != 0).sum(), (t[3] != 0).sum(), a[4].max(), t[4].max(), a[4].mean(), t[4].mean(), dev[ok].max(), np.quantile(dev[ok], .999), (a[4] > t[4]).sum()), flush=True)
