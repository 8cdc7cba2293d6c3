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
if os.environ.get('F20_GP') == '1':                    # BASELINE configs[2]: GP residual in the shooting
    from ad_mpc_amd.config import set_gp
    from ad_mpc_amd.scenarios import grid_gp
    set_gp(cfg, grid_gp())
eng = BatchSolver(cfg, device=0); d = eng.to_device
a = [d(s[k]) for k in ("x0", "yref", "yref_e", "p")]; x0b, u0b = d(s["xbar"]), d(s["ubar"])
cost = torch.empty(B, dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
L = ctypes.CDLL(_lib.LIB_PATH); buf = (ctypes.c_ulonglong * 16)()
ts = []
for rep in range(steps + 2):
    x = x0b.clone(); u = u0b.clone()
    if rep == 2 and hasattr(L, 'admpc_debug_f20_ticks'): L.admpc_debug_f20_ticks(buf)          # clear after the warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.solve(*a, x, u, cost, st, it)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
rc = L.admpc_debug_f20_ticks(buf) if hasattr(L, 'admpc_debug_f20_ticks') else 1
t = np.median(ts[2:]); itc = it.cpu().numpy()
print("B %d: %.3f ms/step (host clock, timers build) %.2f M solves/s; iters mean %.2f max %d" % (B, t * 1e3, B / t / 1e6, itc.mean(), itc.max()))
if rc == 0:
    names = ["ticket+setup", "A1 state RK4 + model", "A2 sensitivity columns", "C condensing", "D trial", "D interior point", "E expand + outputs"]
    v = np.array([buf[i] for i in range(7)], dtype=float) / steps
    tot = v.sum()
    for n, x_ in zip(names, v):
        print("  %-26s %10.0f ticks/step %5.1f %%   %.2f us per instance" % (n, x_, 100 * x_ / tot, x_ / B / 100.0))
    print("  wave-time per step: %.0f ticks = %.1f us x waves; longest wave %.1f us" % (buf[8] / steps, buf[8] / steps / 100.0, buf[9] / 100.0))
# ---- timeline of the last step: start / IPM start / end of every instance (s_memrealtime, 10 ns).  ADMPC_LIB=libadmpc_trace.so
#      (built with -DADMPC_F20_TRACE only) gives it without the phase stamps, i.e. at the shipped kernel's speed.
tr = (ctypes.c_ulonglong * (4 * B))()
if B <= 8192 and hasattr(L, "admpc_debug_f20_trace") and L.admpc_debug_f20_trace(tr, B) == 0:
    a = np.array(tr[:], dtype=np.int64).reshape(B, 4)
    t0 = a[:, 0].min(); st_ = (a[:, 0] - t0) / 100.0; en = (a[:, 1] - t0) / 100.0; mid = np.where(a[:, 3] > 0, (a[:, 3] - t0) / 100.0, np.nan)
    print("  timeline (us): last start %.1f, kernel end %.1f; starts: p50 %.1f p90 %.1f p99 %.1f" % (st_.max(), en.max(), *np.percentile(st_, [50, 90, 99])))
    late = np.argsort(-en)[:10]
    print("  last to finish: " + ", ".join("#%d it %d start %.0f ipm@%.0f end %.0f" % (i, itc[i], st_[i], mid[i], en[i]) for i in late))
    for k in sorted(set(itc)):
        m = itc == k
        per = ((en - mid)[m] / max(k, 1)) if k > 0 else np.zeros(1)
        print("   it %2d: n %4d  start mean %5.0f max %5.0f | A+C+trial mean %5.1f max %5.1f | dur mean %6.1f max %6.1f | end max %6.1f | us/iter mean %.1f max %.1f" % (k, m.sum(), st_[m].mean(), st_[m].max(), np.nanmean((mid - st_)[m]), np.nanmax((mid - st_)[m]), (en - st_)[m].mean(), (en - st_)[m].max(), en[m].max(), per.mean(), per.max()))
    busy = np.zeros(int(en.max()) + 2)
    for s0, e0 in zip(st_, en): busy[int(s0):int(e0) + 1] += 1
    print("  resident instances every 20 us: " + " ".join("%d" % busy[i] for i in range(0, len(busy), 20)))
