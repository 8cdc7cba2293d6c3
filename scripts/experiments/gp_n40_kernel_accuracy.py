"""Which kernel is closer to the truth on the N = 40 + GP family?  Device kernel R and kernel S (ADMPC_QP=seg), and the fp64 oracle, each against the
oracle in 80-bit arithmetic (make -C oracle longdouble) on the same seeded batch.  python3 scripts/experiments/gp_n40_kernel_accuracy.py [B] [seeds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config, set_gp
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios, grid_gp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
o64 = Oracle(omp=True); o80 = Oracle(variant="ld")
for N, gp in ((40, True), (40, False), (20, True)):
    cfg = default_config(N=N)
    if gp: set_gp(cfg, grid_gp())
    for seed in range(seeds):
        s = random_scenarios(B, N=N, seed=100 + seed)
        a = (s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
        t = o80.solve_batch(cfg, *a)
        r = {"oracle fp64": o64.solve_batch(cfg, *a, nthreads=16)}
        for name, qp in (("kernel R", "riccati"), ("kernel S" if N > 20 else "kernel F", "seg" if N > 20 else None)):
            if qp: os.environ["ADMPC_QP"] = qp
            else: os.environ.pop("ADMPC_QP", None)
            r[name] = BatchSolver(cfg, device=0).solve_numpy(*a)
        os.environ.pop("ADMPC_QP", None)
        for name, g in r.items():
            same = (g[3] == t[3]) & (g[4] == t[4]) & (t[3] == 0)
            du = np.abs(g[1] - t[1])[same].max(axis=(1, 2)); dx = np.abs(g[0] - t[0])[same].max(axis=(1, 2))
            print("N %d %s seed %d  %-12s vs 80-bit: same status and iterations %d of %d; |du| median %.1e 99%% %.1e max %.1e; |dx| median %.1e max %.1e"
                  % (N, "GP" if gp else "nominal", 100 + seed, name, same.sum(), B, np.median(du), np.quantile(du, 0.99), du.max(), np.median(dx), dx.max()), flush=True)
