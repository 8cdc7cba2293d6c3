"""Bitwise comparison of two builds of the library over several different batches solved one after another on ONE handle each
(B > the persistent grid, so that the work order -- and with it the expansion queue of admpc_fused20_kernel -- is active):
    python scripts/defer_check.py libadmpc.so libadmpc_nodefer.so [B] [nseeds]
A stale read of a pushed step (the buffers are reused by every launch) would show as a difference in xbar / ubar."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import ad_mpc_amd._lib as _lib
    _lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", sys.argv[2])
    import torch
    from ad_mpc_amd.config import default_config
    from ad_mpc_amd.engine import BatchSolver
    from ad_mpc_amd.scenarios import random_scenarios
    B, nseeds, out = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    cfg = default_config(N=20, Ts=0.05)
    eng = BatchSolver(cfg, device=0)
    d = eng.to_device
    res = {}
    for seed in range(nseeds):
        sc = random_scenarios(B, N=20, Ts=0.05, seed=100 + seed, start=0, blend=(100.0, 110.0) if seed % 2 == 0 else (3.0, 5.0))
        xb, ub = d(sc["xbar"]), d(sc["ubar"])
        cost = torch.zeros(B, dtype=torch.float64, device=xb.device); st = torch.full((B,), -7, dtype=torch.int32, device=xb.device); it = torch.zeros_like(st)
        eng.solve(d(sc["x0"]), d(sc["yref"]), d(sc["yref_e"]), d(sc["p"]), xb, ub, cost, st, it)
        torch.cuda.synchronize()
        res["x%d" % seed] = xb.cpu().numpy(); res["u%d" % seed] = ub.cpu().numpy()
        res["c%d" % seed] = cost.cpu().numpy(); res["s%d" % seed] = st.cpu().numpy(); res["i%d" % seed] = it.cpu().numpy()
    np.savez(out, **res)
    eng.close()
    sys.exit(0)
libs = sys.argv[1:3]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
nseeds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
outs = []
for lib in libs:
    out = "/tmp/defer_check_%s.npz" % lib.replace(".so", "")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", lib, str(B), str(nseeds), out])
    outs.append(np.load(out))
bad = 0
for k in outs[0].files:
    a, b = outs[0][k], outs[1][k]
    same = np.array_equal(a, b)
    if not same:
        bad += 1
        print("DIFFERENT %s: %d entries, max |d| %.3e" % (k, int((a != b).sum()), float(np.nanmax(np.abs(a.astype(float) - b.astype(float))))))
print("defer_check: %d arrays compared (%s vs %s, B=%d, %d batches): %s" % (len(outs[0].files), libs[0], libs[1], B, nseeds, "bit-identical" if bad == 0 else "%d DIFFER" % bad))
sys.exit(1 if bad else 0)
