"""Parity census on the GPU box (uses oracle/ as the checker -- analysis script, not part of the product): the HIP path against the
CPU oracle over many seeded batches, both scenario families, every device path (condensed pipeline, kernel R in one launch and
split in two, fp32).      python3 scripts/gpu_parity_census.py [seeds]   ->  one line per shape"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config, set_gp
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios, grid_gp

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
TIGHT = len(sys.argv) > 2 and sys.argv[2] == "tight"       # the tight stop levels of rounds 1-2 instead of the default (the reference's HPIPM BALANCE levels)
print("interior-point stop levels: %s" % ("tight (1e-10 / 1e-9 / step 1e-6)" if TIGHT else "default = the reference's (1e-8 / 1e-8 / no step test)"), flush=True)
o = Oracle(omp=True)
nth = min(64, os.cpu_count() or 8)
total = 0
for N, B, dt, gp in ((20, 4096, np.float64, False), (20, 4096, np.float64, True), (20, 8192, np.float64, False), (40, 4096, np.float64, False), (40, 4096, np.float64, True),
                     (40, 8192, np.float64, False), (80, 2048, np.float64, False), (80, 8192, np.float64, False), (128, 1024, np.float64, False), (80, 16384, np.float32, False)):
    cfg = default_config(N=N)
    if TIGHT:
        from ad_mpc_amd.config import tight_ipm
        tight_ipm(cfg)
    if gp:
        set_gp(cfg, grid_gp())              # BASELINE configs[2]: GP residual-dynamics correction active
    eng = BatchSolver(cfg, device=0)
    for kw in ({}, {"blend": (3.0, 5.0)}):
        n = 0; bad_status = 0; it_off = 0; it_off2 = 0; du = 0.0; dx = 0.0; du_off = 0.0; mx = 0; t0 = time.time()
        for seed in range(seeds):
            s = random_scenarios(B, N=N, seed=100 + seed, **kw)
            g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], dtype=dt)
            r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=nth)
            n += B; bad_status += int((g[3] != r[3]).sum())
            ok = r[3] == 0
            off = ok & (g[4] != r[4])
            it_off += int(off.sum()); it_off2 += int((ok & (np.abs(g[4] - r[4]) > 1)).sum())
            du = max(du, float(np.abs(g[1][ok] - r[1][ok]).max(initial=0))); dx = max(dx, float(np.abs(g[0][ok] - r[0][ok]).max(initial=0)))
            du_off = max(du_off, float(np.abs(g[1][off] - r[1][off]).max(initial=0)))
            mx = max(mx, int(g[4].max()))
        total += n
        fam = ("dynamic " if kw else "kinematic") + (" + GP residual" if gp else "")
        if dt == np.float64:
            print(f"N {N:3d} B {B:5d} f64 {fam}: {n:6d} instances, status mismatches {bad_status}, iteration counts differing {it_off} (by more than one: {it_off2}; "
                  f"their max |du| {du_off:.1e}), max |du| {du:.2e}, max |dx| {dx:.2e}, max iterations {mx}  [{time.time() - t0:.0f} s]", flush=True)
        else:
            print(f"N {N:3d} B {B:5d} f32 {fam}: {n:6d} instances, status mismatches {bad_status}, max |du| {du:.2e}, max |dx| {dx:.2e} against the fp64 oracle "
                  f"(inputs of size 10; documented bound 2.5e-3), max iterations {mx}  [{time.time() - t0:.0f} s]", flush=True)
    del eng
print("total instances compared:", total)

# ---- the quadrotor path (fast path at N = 10, generic path at N = 5 and 16)
from oracle.quad_oracle import QuadOracle
from ad_mpc_amd.quad_config import default_quad_config
from ad_mpc_amd.quad_scenarios import random_quad_scenarios
from ad_mpc_amd.engine import QuadBatchSolver
qo = QuadOracle()
for N, B in ((10, 4096), (5, 1024), (16, 1024), (20, 2048), (24, 128)):      # N = 20: the segmented two-wave kernel (round 4); 24: the dense two-wave kernel
    qc = default_quad_config(); qc.N = N
    qe = QuadBatchSolver(qc)
    n = 0; bad = 0; itoff = 0; du = 0.0; dx = 0.0; mx = 0
    for seed in range(seeds):
        s = random_quad_scenarios(B, qc, seed=200 + seed)
        g = qe.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
        r = qo.solve_batch(qc, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=nth)
        n += B; bad += int((g[3] != r[3]).sum()); ok = r[3] == 0
        itoff += int((ok & (g[4] != r[4])).sum()); mx = max(mx, int(g[4].max()))
        du = max(du, float(np.abs(g[1][ok] - r[1][ok]).max(initial=0))); dx = max(dx, float(np.abs(g[0][ok] - r[0][ok]).max(initial=0)))
    print(f"quadrotor N {N:2d} B {B:5d}: {n:6d} instances, status mismatches {bad}, iteration counts differing {itoff}, max |du| {du:.2e}, max |dx| {dx:.2e}, max iterations {mx}", flush=True)
    total += n
print("total instances compared (car + quadrotor):", total)
