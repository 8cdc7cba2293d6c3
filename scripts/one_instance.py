"""One instance through another build of the library: python scripts/one_instance.py LIB seed index [gp]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", sys.argv[1])
from ad_mpc_amd.config import default_config, set_gp
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios, grid_gp
cfg = default_config(N=20)
if len(sys.argv) > 4: set_gp(cfg, grid_gp())
s = random_scenarios(1, N=20, seed=int(sys.argv[2]), start=int(sys.argv[3]))
g = BatchSolver(cfg).solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
print("iters", g[4], "status", g[3], "cost %.12g" % g[2][0])
