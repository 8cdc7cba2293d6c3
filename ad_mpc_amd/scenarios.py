"""Seeded synthetic scenario generators for the BASELINE.json configs.

All generators are pure numpy, deterministic per (seed, global instance index) so that instance i is
identical however the batch is sharded over ranks (config 4).
"""
import math

import numpy as np

from .config import NX, NU, NY, BLEND_MIN, BLEND_MAX
from .host import yaw_fix, vel_switch


def straight_scenario(N=20, Ts=0.05, v=5.0):
    """Config 1: single vehicle on a straight path (plumbing case)."""
    x0 = np.array([0.0, 0.0, 0.0, v, 0.0, 0.0, 0.0])
    k = np.arange(N + 1)
    xref = np.zeros((N + 1, NX))
    xref[:, 0] = v * Ts * k
    xref[:, 3] = v
    uref = np.zeros((N, NU))
    return x0, xref, uref


def random_scenarios(B, N=20, Ts=0.05, seed=1234, start=0, blend=(BLEND_MIN, BLEND_MAX), init="x0"):
    """Configs 2-5: random (x0, curved constant-curvature reference) scenarios.

    Returns a dict of float64 arrays: x0 [B,7], yref [B,N,9], yref_e [B,7], p [B], xbar [B,N+1,7],
    ubar [B,N,2], xref [B,N+1,7] (before the yaw fix).
    Instance ``start+i`` only depends on (seed, start+i).
    """
    x0 = np.empty((B, NX)); xref = np.empty((B, N + 1, NX))
    for i in range(B):
        rng = np.random.default_rng([seed, start + i])
        px, py = rng.uniform(-50, 50, 2)
        psi = rng.uniform(-math.pi, math.pi)
        vx = rng.uniform(2, 15); vy = rng.uniform(-0.3, 0.3); r = rng.uniform(-0.3, 0.3); dl = rng.uniform(-0.2, 0.2)
        kappa = rng.uniform(-0.05, 0.05); ey = rng.uniform(-1, 1); epsi = rng.uniform(-0.2, 0.2)
        vref = rng.uniform(3, 15)
        x0[i] = (px, py, psi, vx, vy, r, dl)
        # arc of curvature kappa starting at the pose offset laterally by ey and rotated by epsi
        th0 = psi + epsi
        sx = px - ey * math.sin(psi); sy = py + ey * math.cos(psi)
        s = vref * Ts * np.arange(N + 1)
        th = th0 + kappa * s
        if abs(kappa) > 1e-9:
            xr = sx + (np.sin(th) - math.sin(th0)) / kappa
            yr = sy - (np.cos(th) - math.cos(th0)) / kappa
        else:
            xr = sx + s * math.cos(th0); yr = sy + s * math.sin(th0)
        xref[i] = 0.0
        xref[i, :, 0] = xr; xref[i, :, 1] = yr; xref[i, :, 2] = th; xref[i, :, 3] = vref
    uref = np.zeros((B, N, NU))
    return assemble(x0, xref, uref, blend=blend, init=init)


def assemble(x0, xref, uref, blend=(BLEND_MIN, BLEND_MAX), init="x0"):
    """Turn (x0, xref[N+1], uref[N]) batches into solver inputs exactly as run_optimization does
    (ad_3d_optimizer.py:420-450): yaw fix, yref assembly, blend switch, initial iterate."""
    x0 = np.asarray(x0, dtype=np.float64); xref = np.asarray(xref, dtype=np.float64); uref = np.asarray(uref, dtype=np.float64)
    B, Np1, _ = xref.shape
    N = Np1 - 1
    xr = xref.copy()
    xr[:, :, 2] = yaw_fix(x0[:, 2:3], xr[:, :, 2])
    yref = np.concatenate([xr[:, :N, :], uref], axis=2)
    yref_e = xr[:, N, :].copy()
    p = vel_switch(x0[:, 3], blend[0], blend[1])
    if init == "x0":
        xbar = np.repeat(x0[:, None, :], N + 1, axis=1)
    elif init == "zeros":
        xbar = np.zeros((B, N + 1, NX))
    else:
        raise ValueError(init)
    ubar = np.zeros((B, N, NU))
    return dict(x0=x0, yref=np.ascontiguousarray(yref), yref_e=yref_e, p=p, xbar=xbar, ubar=ubar, xref=xref)


def grid_gp(feat_ranges=((2.0, 15.0), (-0.3, 0.3), (-0.3, 0.3)), M=20, seed=4321):
    """Config 3: three 1-D SE GPs, features v_x, v_y, psi_dot -> outputs dims 3,4,5."""
    rng = np.random.default_rng(seed)
    gps = []
    for g, (lo, hi) in enumerate(feat_ranges):
        Z = np.linspace(lo, hi, M)
        gps.append(dict(feat=3 + g, out=3 + g, Z=Z, alpha=rng.normal(0.0, 0.3, M), length_scale=(hi - lo) / 5.0,
                        sigma_f=1.0, ymean=0.0))
    return gps
