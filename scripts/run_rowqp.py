"""Driver for profiling kernel R: python scripts/run_rowqp.py N B [steps] [dtype] -- repeated solves of one synthetic batch."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("ADMPC_QP", "riccati")
import ad_mpc_amd._lib as _lib
if os.environ.get("ADMPC_LIB"): _lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", os.environ["ADMPC_LIB"])
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
N = int(sys.argv[1]); B = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"; tdt = torch.float32 if f32 else torch.float64
cfg = default_config(N=N)
s = random_scenarios(B, N=N, seed=1234)
eng = BatchSolver(cfg, device=0)
d = (lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=tdt, device="cuda"))
tx0, tyr, tye, tp = d(s["x0"]), d(s["yref"]), d(s["yref_e"]), d(s["p"])
x0b, u0b = d(s["xbar"]), d(s["ubar"])
cost = torch.empty(B, dtype=tdt, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty(B, dtype=torch.int32, device="cuda")
ts = []
for rep in range(steps + 2):
    x = x0b.clone(); u = u0b.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.solve(tx0, tyr, tye, tp, x, u, cost, st, it)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = np.median(ts[2:])
itc = it.cpu().numpy()
print("N %d B %d: %.3f ms/step %.3f M solves/s; iters mean %.2f max %d; quad-max mean %.2f" % (N, B, t * 1e3, B / t / 1e6, itc.mean(), itc.max(), itc[: B // 4 * 4].reshape(-1, 4).max(1).mean()))
if os.environ.get("ADMPC_LIB"):
    import ctypes
    try: ctypes.CDLL(_lib.LIB_PATH).admpc_rowqp_dump_timers()
    except Exception as e: print("no timers:", e)
