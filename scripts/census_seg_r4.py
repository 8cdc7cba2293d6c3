"""Round-4 side census on the GPU box: kernel S on request at N = 60 / 80 (ADMPC_QP=seg), kernel S against kernel R at the tight levels at N = 40,
the quadrotor's segmented kernel.  python3 scripts/census_seg_r4.py [seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config, tight_ipm
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
o = Oracle(omp=True); nth = min(64, os.cpu_count() or 8)
for N, B, tight, qp in ((40, 4096, True, None), (60, 2048, False, "seg"), (80, 2048, False, "seg"), (60, 2048, True, "seg"), (80, 2048, True, "seg")):
    cfg = default_config(N=N)
    if tight: tight_ipm(cfg)
    if qp: os.environ["ADMPC_QP"] = qp
    else: os.environ.pop("ADMPC_QP", None)
    eng = BatchSolver(cfg, device=0)
    for kw in ({}, {"blend": (3.0, 5.0)}):
        n = 0; bad = 0; off = 0; off2 = 0; du = 0.0; dx = 0.0
        for seed in range(seeds):
            s = random_scenarios(B, N=N, seed=300 + seed, **kw)
            g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
            r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=nth)
            n += B; bad += int((g[3] != r[3]).sum()); ok = r[3] == 0
            d = np.abs(g[4] - r[4]); off += int((ok & (d != 0)).sum()); off2 += int((ok & (d > 1)).sum())
            du = max(du, float(np.abs(g[1][ok] - r[1][ok]).max(initial=0))); dx = max(dx, float(np.abs(g[0][ok] - r[0][ok]).max(initial=0)))
        print("kernel S N %d B %d %s %s: %d instances, status mismatches %d, iteration counts differing %d (by more than one: %d), max |du| %.2e, max |dx| %.2e"
              % (N, B, "tight levels" if tight else "default levels", "dynamic" if kw else "kinematic", n, bad, off, off2, du, dx), flush=True)
    del eng
os.environ.pop("ADMPC_QP", None)
