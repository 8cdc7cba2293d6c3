#!/usr/bin/env python3
"""Determinism stress for small batches (waves alone on their SIMDs expose instruction hazards and races that big batches
hide): solve the 6-instance 'active slack / steering bound' set many times, every result must be bit-identical."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
if len(sys.argv) > 2: _lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", sys.argv[2])
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import straight_scenario, assemble, random_scenarios
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cfg = default_config()
x0, xref, uref = straight_scenario(N=cfg.N, Ts=cfg.Ts, v=5.0)
rows = []
for y, d0, v in [(-10.0, 0.5, 3.0), (6.0, 0.5, 14.0), (4.0, -0.5, 6.0), (-6.0, 0.5, 14.0), (0.0, 0.6, 5.0), (0.0, -0.7, 9.0)]:
    x = x0.copy(); x[1] = y; x[6] = d0; x[3] = v; rows.append(x)
X0 = np.array(rows); B = len(rows)
sets = [assemble(X0, np.repeat(xref[None], B, 0), np.repeat(uref[None], B, 0)), random_scenarios(40, N=20, seed=5, blend=(3.0, 5.0)),
        random_scenarios(1, N=20, seed=5, blend=(3.0, 5.0)), random_scenarios(2, N=20, seed=6, blend=(3.0, 5.0))]
eng = BatchSolver(cfg, device=0)
bad = 0
for si, s in enumerate(sets):
    ref = None
    for r in range(reps):
        g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
        if ref is None: ref = g
        else:
            for ai, (a, b) in enumerate(zip(g, ref)):
                if not np.array_equal(a, b, equal_nan=True):
                    print("   array %d (0 x, 1 u, 2 cost, 3 status, 4 iters)" % ai, end=" ")
                    bad += 1
                    d = np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float))
                    print("set %d rep %d differs: max |d| = %g at instance(s) %s" % (si, r, np.nanmax(d), np.unique(np.argwhere(d > 0)[:, 0])[:8]))
                    break
print("stress: %d reps x %d sets, %d mismatching solves" % (reps, len(sets), bad))
sys.exit(1 if bad else 0)
