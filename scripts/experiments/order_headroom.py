"""How much is a perfect work order worth at configs[1]?  Needs ad_mpc_amd/libadmpc_hint.so (make -C ad_mpc_amd/csrc variant TAG=hint EXTRA=-DF20_ORDER_HINT):
the order kernel then bins by a per-instance effort the caller hands in.  Here: the iteration counts of a first solve of the SAME batch (knowledge no
real caller has) -- an upper bound for what any estimate can buy, not a product mode.  python3 scripts/experiments/order_headroom.py [B]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", "libadmpc_hint.so")
import numpy as np, torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = 20
cfg = default_config(N=N); sc = random_scenarios(B, N=N, seed=1234)
eng = BatchSolver(cfg, device=0); d = eng.to_device; lib = eng.lib
lib.admpc_debug_f20_order_hint.argtypes = [C.c_void_p]
x0, yr, ye, p = d(sc["x0"]), d(sc["yref"]), d(sc["yref_e"]), d(sc["p"])
it = torch.zeros(B, dtype=torch.int32, device="cuda:0")
def rate(K=40, W=10):
    xs = [(d(sc["xbar"]), d(sc["ubar"])) for _ in range(K + W)]
    for i in range(W): eng.solve(x0, yr, ye, p, xs[i][0], xs[i][1], None, None, it)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(W, W + K): eng.solve(x0, yr, ye, p, xs[i][0], xs[i][1], None, None, it)
    torch.cuda.synchronize(); return B * K / (time.time() - t0)
r0 = rate()
iters = it.cpu().numpy()
print("estimate of the product: %.2f M solves/s; iterations mean %.2f max %d, %d instances without any" % (r0 / 1e6, iters.mean(), iters.max(), (iters == 0).sum()))
for scale, name in ((5, "bins = 5 x the true iteration count"), (1, "bins = the true iteration count")):
    hint = torch.tensor(np.minimum(63, iters * scale).astype(np.int32), device="cuda:0")
    assert lib.admpc_debug_f20_order_hint(C.c_void_p(hint.data_ptr())) == 0
    print("%s: %.2f M solves/s" % (name, rate() / 1e6))
assert lib.admpc_debug_f20_order_hint(C.c_void_p(0)) == 0
print("back on the estimate: %.2f M solves/s" % (rate() / 1e6))
