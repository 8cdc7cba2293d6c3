"""Ad-hoc first GPU check: shooting + solve parity against the oracle, prints statistics."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle

o = Oracle()
for N, B, blend in [(20, 256, (3, 5)), (20, 256, (100, 110)), (40, 64, (3, 5))]:
    cfg = default_config(N=N)
    s = random_scenarios(B, N=N, blend=blend)
    eng = BatchSolver(cfg)
    # shooting
    phi, A, Bm = eng.shoot(eng.to_device(s["xbar"]), eng.to_device(s["ubar"]), eng.to_device(s["p"]))
    torch.cuda.synchronize()
    phi = phi.cpu().numpy(); A = A.cpu().numpy(); Bm = Bm.cpu().numpy()
    e = 0
    for b in range(min(B, 32)):
        for k in range(N):
            ph, a, bb = o.rk4_sens(cfg, s["xbar"][b, k], s["ubar"][b, k], s["p"][b], cfg.Ts)
            e = max(e, np.abs(ph - phi[b, k]).max(), np.abs(a - A[b, k]).max(), np.abs(bb - Bm[b, k]).max())
    print("N", N, "blend", blend, "shoot max err", e, flush=True)
    t = time.time()
    x, u, cost, st, it = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    dt = time.time() - t
    xo, uo, co, so, io = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    print(" solve: time", dt, "status nz", (st != 0).sum(), "iters gpu mean", it.mean(), "oracle mean", io.mean(), "iters differ", (it != io).sum())
    print(" max|u-uo|", np.abs(u - uo).max(), "max|x-xo|", np.abs(x - xo).max(), "max|cost-co|", np.abs(cost - co).max(), flush=True)
