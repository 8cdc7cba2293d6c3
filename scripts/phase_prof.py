#!/usr/bin/env python3
"""Phase breakdown of the N = 20 interior-point kernel: needs ad_mpc_amd/libadmpc_PROF.so (built with -DADMPC_PHASE_TIMERS)."""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd/libadmpc_PROF.so")      # load the instrumented build instead of the product library
if True:
    import torch
    from ad_mpc_amd.config import default_config
    from ad_mpc_amd.engine import BatchSolver
    from ad_mpc_amd.scenarios import random_scenarios
    cfg = default_config(N=20, Ts=0.05)
    sc = random_scenarios(4096, N=20, Ts=0.05, seed=1234, start=0, blend=(100.0, 110.0))
    if len(sys.argv) > 1:          # e.g. "1920,4063,2920": only these instances (waves alone on their SIMDs)
        idx = [int(v) for v in sys.argv[1].split(",")]
        sc = {k: v[idx] for k, v in sc.items()}
    eng = BatchSolver(cfg, device=0)
    d = eng.to_device
    for _ in range(3):
        xb, ub = d(sc["xbar"]), d(sc["ubar"])
        eng.solve(d(sc["x0"]), d(sc["yref"]), d(sc["yref_e"]), d(sc["p"]), xb, ub)
    torch.cuda.synchronize()
    eng.close()
