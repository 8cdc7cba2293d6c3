"""Phase split of the fused N = 20 kernel: python scripts/f20_phases.py [B] [steps]  (needs `make -C ad_mpc_amd/csrc timers`).
Prints the wave-time each phase takes, summed over all waves of a step (s_memtime ticks, 100 MHz), and the step time."""
import os, sys, time, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", os.environ.get("ADMPC_LIB", "libadmpc_timers.so"))
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = default_config(N=20); s = random_scenarios(B, N=20, seed=1234)
eng = BatchSolver(cfg, device=0); d = eng.to_device
a = [d(s[k]) for k in ("x0", "yref", "yref_e", "p")]; x0b, u0b = d(s["xbar"]), d(s["ubar"])
cost = torch.empty(B, dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
L = ctypes.CDLL(_lib.LIB_PATH); buf = (ctypes.c_ulonglong * 16)()
ts = []
for rep in range(steps + 2):
    x = x0b.clone(); u = u0b.clone()
    if rep == 2: L.admpc_debug_f20_ticks(buf)          # clear after the warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.solve(*a, x, u, cost, st, it)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
rc = L.admpc_debug_f20_ticks(buf)
t = np.median(ts[2:]); itc = it.cpu().numpy()
print("B %d: %.3f ms/step (host clock, timers build) %.2f M solves/s; iters mean %.2f max %d" % (B, t * 1e3, B / t / 1e6, itc.mean(), itc.max()))
if rc == 0:
    names = ["ticket+setup", "A1 state RK4 + model", "A2 sensitivity columns", "C condensing", "D trial", "D interior point", "E expand + outputs"]
    v = np.array([buf[i] for i in range(7)], dtype=float) / steps
    tot = v.sum()
    for n, x_ in zip(names, v):
        print("  %-26s %10.0f ticks/step %5.1f %%   %.2f us per instance" % (n, x_, 100 * x_ / tot, x_ / B / 100.0))
    print("  wave-time per step: %.0f ticks = %.1f us x waves; longest wave %.1f us" % (buf[8] / steps, buf[8] / steps / 100.0, buf[9] / 100.0))
