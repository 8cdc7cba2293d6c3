"""Throughput of the quadrotor path (not a BASELINE config): python scripts/bench_quad.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import ad_mpc_amd._lib as _lib
if os.environ.get("ADMPC_LIB"): _lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", os.environ["ADMPC_LIB"])
from ad_mpc_amd.quad_config import default_quad_config
from ad_mpc_amd.quad_scenarios import random_quad_scenarios
from ad_mpc_amd.engine import QuadBatchSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 10                 # horizon: 10 (the shipped code), 20 (the class default)
cfg = default_quad_config(N=NQ, t_horizon=0.1 * NQ); s = random_quad_scenarios(B, cfg, seed=1)
if os.environ.get("QUAD_ITMAX"): cfg.ipm_iter_max = int(os.environ["QUAD_ITMAX"])      # 0: shooting + condensing + expansion only (cost split)
eng = QuadBatchSolver(cfg); d = lambda a: torch.as_tensor(a, device="cuda")
x0, yr, ye, xb0, ub0 = d(s["x0"]), d(s["yref"]), d(s["yref_e"]), d(s["xbar"]), d(s["ubar"])
it = torch.empty(B, dtype=torch.int32, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); co = torch.empty(B, dtype=torch.float64, device="cuda")
ts = []
for rep in range(8):
    xb, ub = xb0.clone(), ub0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.solve(x0, yr, ye, xb, ub, co, st, it); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = np.median(ts[2:])
print("quadrotor nx=13 nu=4 N=%d B=%d: %.3f ms/step, %.2f M solves/s; IPM iterations mean %.2f max %d; status != 0: %d" % (cfg.N, B, t * 1e3, B / t / 1e6, it.float().mean().item(), it.max().item(), int((st != 0).sum())))
ts = []
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.shoot(xb0, ub0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("  shooting alone (admpc_quad_shoot_batch, includes writing A, B, phi to global memory): %.3f ms" % (np.median(ts[2:]) * 1e3))
try:
    import ctypes
    ctypes.CDLL(eng.lib._name).admpc_quad_dump_timers()       # only in a -DADMPC_QUAD_TIMERS build
except Exception:
    pass
from oracle.quad_oracle import QuadOracle   # CPU baseline of the same step (analysis script)
o = QuadOracle(); n = min(B, 1024); t0 = time.perf_counter()
o.solve_batch(cfg, s["x0"][:n], s["yref"][:n], s["yref_e"][:n], s["xbar"][:n], s["ubar"][:n], nthreads=os.cpu_count())
print("CPU oracle, %d threads: %.0f solves/s" % (os.cpu_count(), n / (time.perf_counter() - t0)))
