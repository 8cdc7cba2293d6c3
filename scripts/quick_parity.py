import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join('/root/repo/ad_mpc_amd', sys.argv[1])
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
cfg = default_config(N=20); s = random_scenarios(777, N=20, seed=5, blend=(3.0, 5.0))
g = BatchSolver(cfg).solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
o = Oracle(omp=True).solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=16)
print(sys.argv[1], "status eq", (g[3] == o[3]).all(), "iters eq", (g[4] == o[4]).all(), "max|du| %.2e max|dx| %.2e cost rel %.2e" % (np.abs(g[1]-o[1]).max(), np.abs(g[0]-o[0]).max(), np.abs(g[2]/o[2]-1).max()))
