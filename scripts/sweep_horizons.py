"""One-off confidence sweep of the Riccati path across its instantiation boundaries: GPU vs oracle (needs the GPU)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
o = Oracle(omp=True)
worst = 0.0
for N in (2, 3, 7, 19, 21, 27, 28, 32, 33, 45, 46, 64, 65, 96, 97, 128):
    B = 1200 if N <= 46 else (300 if N <= 65 else 96)
    cfg = default_config(N=N)
    s = random_scenarios(B, N=N, seed=1000 + N, blend=(3.0, 5.0))
    eng = BatchSolver(cfg, device=0)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    g2 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=16)
    same = (g[4] == r[4]); ok = (g[3] == 0) & (r[3] == 0)
    du = np.abs(g[1] - r[1]).reshape(B, -1).max(1)
    print("N %3d B %4d status equal %s  iters equal %.4f  max|du| same-iters %.2e  overall %.2e  repeatable %s  mean iters %.2f" %
          (N, B, (g[3] == r[3]).all(), same.mean(), du[same & ok].max(initial=0), du[ok].max(initial=0), all((a == b).all() for a, b in zip(g, g2)), g[4].mean()), flush=True)
    worst = max(worst, du[same & ok].max(initial=0))
    eng.close()
print("worst same-iteration deviation %.2e" % worst)
