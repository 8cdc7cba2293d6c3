import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
o = Oracle(omp=True); cfg = default_config(N=80); eng = BatchSolver(cfg)
for kw in ({}, {"blend": (3.0, 5.0)}):
    du = 0
    for seed in range(3):
        s = random_scenarios(16384, N=80, seed=100 + seed, **kw)
        g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], dtype=np.float32)
        r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=64)
        du = max(du, np.abs(g[1] - r[1]).max()); assert (g[3] == r[3]).all()
    print("f32 census N 80 B 16384", kw, "max|du| %.3e" % du, "max iters", g[4].max())
