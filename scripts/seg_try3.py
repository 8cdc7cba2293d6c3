import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
orc = Oracle(omp=True)
cfg = default_config(N=40)
s = random_scenarios(130, N=40, seed=9, blend=(3.0, 5.0))
o1 = orc.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
s2 = dict(s); s2["xbar"] = o1[0].copy(); s2["ubar"] = o1[1].copy()
def run(idx):
    t = {k: np.ascontiguousarray(v[idx]) for k, v in s2.items()}
    g = BatchSolver(cfg).solve_numpy(t["x0"], t["yref"], t["yref_e"], t["p"], t["xbar"], t["ubar"])
    o = orc.solve_batch(cfg, t["x0"], t["yref"], t["yref_e"], t["p"], t["xbar"], t["ubar"], nthreads=8)
    eu = np.abs(g[1] - o[1]).reshape(len(idx), -1).max(1); ex = np.abs(g[0] - o[0]).reshape(len(idx), -1).max(1)
    bad = [(int(idx[i]), int(g[3][i]), int(o[3][i]), int(g[4][i]), int(o[4][i]), float(eu[i]), float(ex[i])) for i in range(len(idx)) if not (eu[i] < 1e-7 and ex[i] < 1e-6)]
    return bad
allb = run(np.arange(130))
print("full batch: %d bad" % len(allb))
for b in allb[:20]: print("  inst %d status %d/%d iters %d/%d du %.2e dx %.2e" % b)
for sub in ([69], [68, 69], list(range(60, 70)), list(range(0, 70)), [b[0] for b in allb[:8]]):
    r = run(np.array(sub))
    print("subset", sub[:12], "...", len(sub), "->", [(b[0], "%.1e" % b[5]) for b in r][:10])
