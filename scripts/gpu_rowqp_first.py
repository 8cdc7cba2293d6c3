"""First GPU check of kernel R (row-mapped Riccati IPM): parity with the oracle, repeatability, timing."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
o = Oracle(omp=True)
os.environ["ADMPC_QP"] = "riccati"
for N, B in ((20, 64), (5, 8), (2, 4), (40, 64), (20, 1024), (40, 512), (80, 64), (128, 16), (33, 7)):
    cfg = default_config(N=N)
    s = random_scenarios(B, N=N, seed=1234, blend=(3.0, 5.0))
    eng = BatchSolver(cfg, device=0)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    g2 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=16)
    ok = (g[3] == 0) & (r[3] == 0)
    print("N %3d B %4d status eq %s iters eq %d/%d max|du| %.2e max|dx| %.2e cost %.2e repeatable %s mean it %.2f max it %d" %
          (N, B, (g[3] == r[3]).all(), (g[4] == r[4]).sum(), B, np.abs(g[1] - r[1])[ok].max(initial=0), np.abs(g[0] - r[0])[ok].max(initial=0),
           np.abs(g[2] - r[2])[ok].max(initial=0), all(np.array_equal(a, b) for a, b in zip(g, g2)), g[4].mean(), g[4].max()), flush=True)
    eng.close()
# timing
for N, B, qp in ((20, 4096, "riccati"), (20, 4096, "dense"), (40, 4096, "riccati"), (80, 2048, "riccati"), (20, 65536, "riccati"), (40, 16384, "riccati")):
    if qp == "riccati": os.environ["ADMPC_QP"] = "riccati"
    else: os.environ.pop("ADMPC_QP", None)
    cfg = default_config(N=N)
    s = random_scenarios(B, N=N, seed=1234)
    eng = BatchSolver(cfg, device=0)
    d = eng.to_device
    tx0, tyr, tye, tp = d(s["x0"]), d(s["yref"]), d(s["yref_e"]), d(s["p"])
    x0b, u0b = d(s["xbar"]), d(s["ubar"])
    cost = torch.empty(B, dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty(B, dtype=torch.int32, device="cuda")
    ts = []
    for rep in range(8):
        x = x0b.clone(); u = u0b.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.solve(tx0, tyr, tye, tp, x, u, cost, st, it)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = np.median(ts[2:])
    print("timing N %3d B %6d %-8s %.3f ms/step  %.2f M solves/s  (mean iters %.2f)" % (N, B, qp, t * 1e3, B / t / 1e6, it.float().mean().item()), flush=True)
    eng.close()
