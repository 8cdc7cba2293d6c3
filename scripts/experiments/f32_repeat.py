"""fp32 path at configs[4] size: is a solve bit-repeatable, and where is the worst instance against the fp64 oracle?  python3 scripts/experiments/f32_repeat.py [lib] [seeds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", sys.argv[1])
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
N, B = 80, 16384
cfg = default_config(N=N); o = Oracle(omp=True); eng = BatchSolver(cfg, device=0)
for kw in ({}, {"blend": (3.0, 5.0)}):
    for seed in range(seeds):
        s = random_scenarios(B, N=N, seed=100 + seed, **kw)
        a = (s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
        g1 = eng.solve_numpy(*a, dtype=np.float32); g2 = eng.solve_numpy(*a, dtype=np.float32); g3 = eng.solve_numpy(*a, dtype=np.float32)
        rep = int((g1[1] != g2[1]).any(axis=(1, 2)).sum()) + int((g1[1] != g3[1]).any(axis=(1, 2)).sum())
        r = o.solve_batch(cfg, *a, nthreads=64)
        du = np.abs(g1[1].astype(np.float64) - r[1]).max(axis=(1, 2)); w = int(np.argmax(du))
        print("%s seed %d: instances differing between three solves of the same batch: %d; worst |du| %.3e at instance %d (device iterations %d, oracle %d, device status %d); second worst %.3e"
              % ("dynamic" if kw else "kinematic", 100 + seed, rep, du[w], w, g1[4][w], r[4][w], g1[3][w], np.sort(du)[-2]), flush=True)
