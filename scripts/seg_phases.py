#!/usr/bin/env python3
"""Phase breakdown of the segmented kernel: needs ad_mpc_amd/libadmpc_timers.so (make -C ad_mpc_amd/csrc timers).
python scripts/seg_phases.py [N B [lib]]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", sys.argv[3] if len(sys.argv) > 3 else "libadmpc_timers.so")
import torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
cfg = default_config(N=N, Ts=0.05)
sc = random_scenarios(B, N=N, Ts=0.05, seed=1234)
eng = BatchSolver(cfg, device=0)
d = eng.to_device
lib = eng.lib
lib.admpc_debug_seg_ticks.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
x0, yr, ye, p = d(sc["x0"]), d(sc["yref"]), d(sc["yref_e"]), d(sc["p"])
for rep in range(3):
    xb, ub = d(sc["xbar"]), d(sc["ubar"])
    torch.cuda.synchronize(); t0 = time.time()
    eng.solve(x0, yr, ye, p, xb, ub)
    torch.cuda.synchronize(); t1 = time.time()
    rc = lib.admpc_debug_seg_ticks(buf)
names = ["ticket", "phase A", "phase C", "trial", "iter top", "factorise+schur", "fwd subst", "iface wait/run", "bwd subst", "expand/step", "cut states", "phase A'", "phase E", "cut operators"]
tot = sum(buf[i] for i in range(14))
print("N %d B %d: step %.3f ms (host timed), timers rc %d, wave-time total %.1f ms (100 MHz ticks)" % (N, B, 1e3 * (t1 - t0), rc, tot / 1e5))
for i in range(14):
    print("  %-18s %6.2f %%   %8.1f us per instance-wave" % (names[i], 100.0 * buf[i] / max(tot, 1), buf[i] / 100.0 / (B * (N // 20))))

# ---- timeline of the last launch: when the instances start and end, by iteration count
import numpy as np
lib.admpc_debug_seg_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
nb = min(B, 16384)
tb = (C.c_ulonglong * (4 * nb))()
if lib.admpc_debug_seg_trace(tb, nb) == 0:
    t = np.frombuffer(tb, dtype=np.uint64).reshape(nb, 4).astype(np.float64)
    it = eng.last_iters.cpu().numpy()[:nb] if hasattr(eng, "last_iters") else None
    t0 = t[:, 0].min(); st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0; dur = en - st
    print("timeline (us): kernel span %.1f; last start %.1f; instance duration min/mean/max %.1f / %.1f / %.1f" % (en.max(), st.max(), dur.min(), dur.mean(), dur.max()))
    order = np.argsort(-en)[:12]
    print("last to finish (inst: start -> end, duration, workgroup):")
    for i in order: print("   %5d: %7.1f -> %7.1f  %6.1f  wg %d" % (i, st[i], en[i], dur[i], int(t[i, 2])))
    edges = np.arange(0, en.max() + 50, 50)
    busy = [(np.minimum(en, b + 50) - np.maximum(st, b)).clip(0).sum() / 50.0 for b in edges]
    print("resident workgroups per 50 us bin:", " ".join("%d" % round(x) for x in busy))
