import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
N, B = 40, 64
cfg = default_config(N=N)
s = random_scenarios(B, N=N, seed=1234, blend=(3.0, 5.0))
eng = BatchSolver(cfg)
xo, uo, co, so, io = Oracle().solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
for rep in range(3):
    x, u, c, st, it = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    err = np.abs(u - uo).reshape(B, -1).max(1)
    bad = np.where(err > 1e-8)[0]
    print("rep", rep, "bad instances", bad.tolist(), "iters gpu", it[bad].tolist(), "oracle", io[bad].tolist(), "err", ["%.1e" % e for e in err[bad]])
    # where inside the horizon is the error?
    for b in bad[:2]:
        e = np.abs(u[b] - uo[b]).max(1)
        print("   inst", b, "stage errors", ["%.0e" % v for v in e])
