#!/usr/bin/env python3
"""Single-vehicle latency (BASELINE configs[0]): one SQP-RTI solve, B = 1, N = 20 -- device-resident call and the full
reference-shaped Python path (AD3DMPC.set_reference + optimize, host arrays in and out)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import straight_scenario, assemble, random_scenarios

cfg = default_config(N=20)
eng = BatchSolver(cfg, device=0)
for name, s in (("straight path (config 1)", None), ("random scenario with active bounds", random_scenarios(64, N=20, seed=1234)),):
    if s is None:
        x0, xref, uref = straight_scenario(N=20, Ts=0.05, v=5.0)
        s = assemble(x0[None], xref[None], uref[None])
        idx = 0
    else:
        idx = 5
    d = eng.to_device
    args = [d(s[k][idx:idx + 1]) for k in ("x0", "yref", "yref_e", "p")]
    xb0, ub0 = d(s["xbar"][idx:idx + 1]), d(s["ubar"][idx:idx + 1])
    cost = torch.empty(1, dtype=torch.float64, device=eng.device); st = torch.empty(1, dtype=torch.int32, device=eng.device); it = torch.empty_like(st)
    ts = []
    for r in range(300):
        xb, ub = xb0.clone(), ub0.clone()
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.solve(*args, xb, ub, cost, st, it)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ts = np.array(ts[50:]) * 1e6
    print("%-38s device-resident solve: median %.1f us, p95 %.1f us, %d IPM iterations" % (name, np.median(ts), np.percentile(ts, 95), int(it.item())))

from ad_mpc_amd.ad_3d import AD3D
from ad_mpc_amd.ad_3d_mpc import AD3DMPC
car = AD3D()
mpc = AD3DMPC(car, t_horizon=1.0, n_nodes=20)
x0, xref, uref = straight_scenario(N=20, Ts=0.05, v=5.0)
car.set_state(list(x0))
ts = []
for r in range(300):
    t = time.perf_counter()
    mpc.set_reference(xref, uref)
    w, x, status = mpc.optimize(return_x=True)
    ts.append(time.perf_counter() - t)
ts = np.array(ts[50:]) * 1e6
print("AD3DMPC.set_reference + optimize (host arrays in/out, reference-shaped API): median %.1f us, p95 %.1f us, status %d" % (np.median(ts), np.percentile(ts, 95), status))
