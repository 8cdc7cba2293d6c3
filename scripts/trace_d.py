"""Per-instance start / duration / block of the interior-point kernel.  Needs the debug build, which packs them into `cost`:
  cd ad_mpc_amd/csrc && hipcc --offload-arch=gfx950 -DADMPC_WSYNC_FENCE_ONLY -DADMPC_TRACE_SCHED -O3 -std=c++17 -fPIC -shared \
      -Wno-unused-function -o ../libadmpc_DBG.so admpc_kernels.hip
Round-1 finding (B = 4096): the first wave dispatched to a SIMD (blocks 0..1023) iterates at 11.1-11.9 us, the second one
(blocks 1024..2047) at 14-18 us while both are busy; `s_setprio 3` on the second wave inverts that."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd/libadmpc_DBG.so")
import torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
cfg = default_config(N=20); sc = random_scenarios(4096, N=20, seed=1234)
eng = BatchSolver(cfg, device=0); d = eng.to_device
for rep in range(3):
    cost = torch.zeros(4096, dtype=torch.float64, device=eng.device); it = torch.zeros(4096, dtype=torch.int32, device=eng.device)
    eng.solve(d(sc["x0"]), d(sc["yref"]), d(sc["yref_e"]), d(sc["p"]), d(sc["xbar"]), d(sc["ubar"]), cost, None, it)
    torch.cuda.synchronize()
pk = cost.cpu().numpy().astype(np.uint64)
t0 = ((pk >> np.uint64(33)) & np.uint64(0xfffff)).astype(np.int64); dur = ((pk >> np.uint64(18)) & np.uint64(0x7fff)).astype(np.int64)
blk = ((pk >> np.uint64(6)) & np.uint64(0xfff)).astype(np.int64); its = (pk & np.uint64(63)).astype(np.int64)
t0 = (t0 - t0.min()) % (1 << 20)
print("kernel span %.1f us" % ((t0 + dur).max() / 100.0))
order = np.argsort(-its)
print(" inst  its  start_us  dur_us  end_us  block  secondary  us/iter")
for i in order[:25]:
    print("%5d %4d %9.1f %7.1f %7.1f %6d %6d %9.2f" % (i, its[i], t0[i] / 100, dur[i] / 100, (t0[i] + dur[i]) / 100, blk[i], blk[i] >= 1024, (dur[i] / 100 - 9) / max(its[i], 1)))
for lo, hi in ((5, 6), (7, 8), (9, 20)):
    m = (its >= lo) & (its <= hi)
    for sec in (0, 1):
        mm = m & ((blk >= 1024) == sec) & (t0 < 300)
        if mm.any(): print("its %d-%d first-ticket %s: n %d mean us/iter %.2f" % (lo, hi, "secondary" if sec else "primary", mm.sum(), ((dur[mm] / 100 - 9) / its[mm]).mean()))
late = (t0 > 300) & (its > 0)
print("IPM instances started late (not a first ticket):", late.sum(), " their mean us/iter %.2f" % (((dur[late] / 100 - 9) / its[late]).mean() if late.any() else 0))
