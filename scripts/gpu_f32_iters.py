import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
N, B = 80, 16384
cfg = default_config(N=N)
s = random_scenarios(B, N=N, seed=1234)
eng = BatchSolver(cfg, device=0)
g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], dtype=np.float32)
it = g[4]
print("hist", np.bincount(it)[:60])
bad = np.nonzero(it >= 30)[0]
print("slow", bad, it[bad])
o = Oracle(omp=True)
sub = bad[:16]
r = o.solve_batch(cfg, s["x0"][sub], s["yref"][sub], s["yref_e"][sub], s["p"][sub], s["xbar"][sub], s["ubar"][sub], nthreads=8)
print("oracle iters", r[4], "err", np.abs(g[1][sub] - r[1]).reshape(len(sub), -1).max(1))
np.save(os.path.join(ROOT, "gpurun_out", "f32_slow_idx.npy"), bad)
