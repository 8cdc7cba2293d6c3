"""python scripts/ric_variants.py libadmpc_X.so: correctness (vs oracle, repeatability) and step time of the Riccati path at
N = 24, 40, 80 for another build of the library in ad_mpc_amd/ (used to compare optimisation levels of that translation unit)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, 'ad_mpc_amd', sys.argv[1])
import torch, time
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
o = Oracle(omp=True)
for N, B in ((24, 512), (33, 512), (40, 2048), (45, 512), (64, 256), (80, 2048)):
    cfg = default_config(N=N); s = random_scenarios(B, N=N, seed=1234, blend=(3.0, 5.0))
    eng = BatchSolver(cfg, device=0); d = eng.to_device
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    g2 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    n = 256
    r = o.solve_batch(cfg, s["x0"][:n], s["yref"][:n], s["yref_e"][:n], s["p"][:n], s["xbar"][:n], s["ubar"][:n], nthreads=16)
    args = [d(s[k]) for k in ("x0", "yref", "yref_e", "p")]
    xb = [d(s["xbar"]) for _ in range(6)]; ub = [d(s["ubar"]) for _ in range(6)]
    eng.solve(*args, xb[0], ub[0]); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(1, 6): eng.solve(*args, xb[i], ub[i])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(sys.argv[1], "N", N, "ms %.3f" % (dt * 1e3), "iters equal", (g[4][:n] == r[4]).mean(), "max|du| %.2e" % np.abs(g[1][:n] - r[1]).max(),
          "repeatable", all((a == b).all() for a, b in zip(g, g2)), flush=True)
    eng.close()
