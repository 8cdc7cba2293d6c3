"""Find instances whose device result differs from the oracle's (analysis script): python scripts/find_mismatch.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config, set_gp
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios, grid_gp
o = Oracle(omp=True)
cfg = default_config(N=20); set_gp(cfg, grid_gp())
eng = BatchSolver(cfg)
for seed in (100, 101, 102):
    s = random_scenarios(4096, N=20, seed=seed)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=32)
    bad = np.nonzero((g[4] != r[4]) | (np.abs(g[1] - r[1]).reshape(4096, -1).max(1) > 1e-8))[0]
    for i in bad:
        print("seed", seed, "instance", i, "device iters", g[4][i], "oracle iters", r[4][i], "status", g[3][i], r[3][i], "max|du| %.3e" % np.abs(g[1][i] - r[1][i]).max(),
              "cost dev %.12g oracle %.12g" % (g[2][i], r[2][i]))
        os.environ["ADMPC_QP"] = "riccati"
        e2 = BatchSolver(cfg)
        h = e2.solve_numpy(s["x0"][i:i+1], s["yref"][i:i+1], s["yref_e"][i:i+1], s["p"][i:i+1], s["xbar"][i:i+1], s["ubar"][i:i+1])
        print("   kernel R on the same instance: iters", h[4][0], "max|du - oracle| %.3e  max|du - fused| %.3e cost %.12g" % (np.abs(h[1][0] - r[1][i]).max(), np.abs(h[1][0] - g[1][i]).max(), h[2][0]))
        del os.environ["ADMPC_QP"]
