"""One-off confidence sweep of the N = 20 path over batch sizes around the grid / SIMD-count boundaries: GPU vs oracle."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ad_mpc_amd.config import default_config, set_gp
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios, grid_gp
from oracle.oracle import Oracle
o = Oracle(omp=True)
cfg = default_config(N=20)
eng = BatchSolver(cfg, device=0)            # ONE engine through all sizes (workspace grows and is reused)
worst = 0.0
for B in (1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 6000, 8193):
    s = random_scenarios(B, N=20, seed=500 + B, blend=(3.0, 5.0) if B % 2 else (100.0, 110.0))
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    g2 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=16)
    same = g[4] == r[4]; ok = (g[3] == 0) & (r[3] == 0)
    du = np.abs(g[1] - r[1]).reshape(B, -1).max(1)
    print("B %5d status equal %s  iters equal %.4f (max diff %d)  max|du| same-iters %.2e  overall %.2e  repeatable %s" %
          (B, (g[3] == r[3]).all(), same.mean(), np.abs(g[4] - r[4]).max(), du[same & ok].max(initial=0), du[ok].max(initial=0),
           all((a == b).all() for a, b in zip(g, g2))), flush=True)
    worst = max(worst, du[same & ok].max(initial=0))
print("worst same-iteration deviation %.2e" % worst)
