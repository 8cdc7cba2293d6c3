import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
orc = Oracle(omp=True)
def cmp(tag, cfg, s):
    g = BatchSolver(cfg).solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    o = orc.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
    ok = o[3] == 0
    eu = np.abs(g[1] - o[1]).reshape(len(ok), -1).max(1); ex = np.abs(g[0] - o[0]).reshape(len(ok), -1).max(1)
    print(tag, "status eq", (g[3] == o[3]).all(), "iters eq", (g[4][ok] == o[4][ok]).all(), "max du %.2e dx %.2e" % (eu[ok].max(), ex[ok].max()), "worst inst", int(np.argmax(np.where(ok, eu, 0))), flush=True)
    return g, o
s = random_scenarios(130, N=40, seed=9, blend=(3.0, 5.0))
g1, o1 = cmp("1 pass          ", default_config(N=40), s)
s2 = dict(s); s2["xbar"] = o1[0].copy(); s2["ubar"] = o1[1].copy()
cmp("1 pass from iterate", default_config(N=40), s2)
cmp("2 passes        ", default_config(N=40, sqp_iters=2), s)
cmp("3 passes        ", default_config(N=40, sqp_iters=3), s)
s3 = {k: v.copy() for k, v in s.items()}; s3["x0"][17, 3] = np.nan; s3["yref"][101, 5, 0] = np.inf
cmp("1 pass, failures", default_config(N=40), s3)
cmp("3 passes, failures", default_config(N=40, sqp_iters=3), s3)
