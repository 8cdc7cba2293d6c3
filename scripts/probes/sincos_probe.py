#!/usr/bin/env python3
"""Accuracy of the model's sin/cos on the device: shoot a batch whose only varying input is the yaw angle and compare
phi[0] = x + h * RK4 increments against the oracle (libm).  Reports the worst relative deviation of the shooting result."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from oracle.oracle import Oracle
cfg = default_config(N=2); eng = BatchSolver(cfg, device=0); o = Oracle()
rng = np.random.default_rng(0); B = 20000
x = np.zeros((B, 3, 7)); u = np.zeros((B, 2, 2)); p = rng.choice([0.0, 0.4, 1.0], B)
x[:, :, 2] = rng.uniform(-40, 40, B)[:, None]; x[:, :, 3] = rng.uniform(2, 15, B)[:, None]; x[:, :, 4] = rng.uniform(-.3, .3, B)[:, None]
x[:, :, 5] = rng.uniform(-.3, .3, B)[:, None]; x[:, :, 6] = rng.uniform(-.6, .6, B)[:, None]; u[:, :, 0] = rng.uniform(-5, 5, B)[:, None]; u[:, :, 1] = rng.uniform(-2, 2, B)[:, None]
phi, A, Bm = eng.shoot(eng.to_device(x), eng.to_device(u), eng.to_device(p)); torch.cuda.synchronize()
phi, A = phi.cpu().numpy(), A.cpu().numpy()
worst = 0.0; worstA = 0.0
for b in range(0, B, 7):
    ph, Ao, Bo = o.rk4_sens(cfg, x[b, 0], u[b, 0], p[b], cfg.Ts)
    worst = max(worst, np.abs(phi[b, 0] - ph).max() / max(1.0, np.abs(ph).max())); worstA = max(worstA, np.abs(A[b, 0] - Ao).max())
print("sincos probe: worst relative deviation of phi %.2e, of A %.2e (|psi| up to 40 rad)" % (worst, worstA))
