"""First GPU check of the fp32 path (config 5): against the fp64 oracle."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
o = Oracle(omp=True)
for N, B, blend in ((20, 256, (100., 110.)), (20, 256, (3., 5.)), (40, 256, (100., 110.)), (80, 512, (100., 110.)), (80, 256, (3., 5.)), (5, 16, (3., 5.))):
    cfg = default_config(N=N)
    s = random_scenarios(B, N=N, seed=1234, blend=blend)
    eng = BatchSolver(cfg, device=0)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], dtype=np.float32)
    g2 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], dtype=np.float32)
    r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=16)
    ok = (g[3] == 0) & (r[3] == 0)
    eu = np.abs(g[1] - r[1]).reshape(B, -1).max(1); ex = np.abs(g[0] - r[0]).reshape(B, -1).max(1)
    print("f32 N %3d B %4d blend %s: status eq %s  max|du| %.2e p99 %.2e  max|dx| %.2e  rel cost %.2e  repeatable %s  iters mean %.2f max %d (oracle %.2f)" %
          (N, B, blend[0], (g[3] == r[3]).all(), eu[ok].max(), np.quantile(eu[ok], 0.99), ex[ok].max(), (np.abs(g[2] - r[2]) / np.abs(r[2]))[ok].max(),
           all(np.array_equal(a, b) for a, b in zip(g, g2)), g[4].mean(), g[4].max(), r[4].mean()), flush=True)
    eng.close()
for N, B in ((80, 16384), (20, 4096), (40, 4096)):
    cfg = default_config(N=N)
    s = random_scenarios(B, N=N, seed=1234)
    eng = BatchSolver(cfg, device=0)
    d = lambda a: eng.to_device(a, torch.float32)
    tx0, tyr, tye, tp = d(s["x0"]), d(s["yref"]), d(s["yref_e"]), d(s["p"])
    x0b, u0b = d(s["xbar"]), d(s["ubar"])
    cost = torch.empty(B, dtype=torch.float32, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty(B, dtype=torch.int32, device="cuda")
    ts = []
    for rep in range(6):
        x = x0b.clone(); u = u0b.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.solve(tx0, tyr, tye, tp, x, u, cost, st, it)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = np.median(ts[2:])
    print("timing f32 N %3d B %6d %.3f ms/step  %.2f M solves/s  (mean iters %.2f max %d, status!=0: %d)" % (N, B, t * 1e3, B / t / 1e6, it.float().mean().item(), it.max().item(), (st != 0).sum().item()), flush=True)
    eng.close()
