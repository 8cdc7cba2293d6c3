#!/usr/bin/env python3
"""Single-vehicle latency of the quadrotor path through the reference-shaped class: Quad3DOptimizer.set_reference_trajectory + run_optimization
(host arrays in and out), N = 10 (the shipped horizon) and N = 20 (the class default)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ad_mpc_amd.quad_3d_optimizer import Quad3DOptimizer
for N, T in ((10, 1.0), (20, 2.0)):
    opt = Quad3DOptimizer(t_horizon=T, n_nodes=N)
    t = np.linspace(0, T, N + 1)
    xt = [np.c_[0.5 * t, 0.2 * t, 1.0 + 0 * t], np.tile([1.0, 0, 0, 0], (N + 1, 1)), np.tile([0.5, 0.2, 0.0], (N + 1, 1)), np.zeros((N + 1, 3))]
    ut = np.full((N + 1, 4), 0.25)
    x0 = [0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    ts = []
    for r in range(300):
        t0 = time.perf_counter()
        opt.set_reference_trajectory(xt, ut)
        w = opt.run_optimization(initial_state=x0)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[50:]) * 1e6
    print("quadrotor N = %d: set_reference_trajectory + run_optimization: median %.1f us, p95 %.1f us, status %d" % (N, np.median(ts), np.percentile(ts, 95), opt.status))
