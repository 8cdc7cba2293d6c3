"""Synthetic quadrotor scenarios: random states around hover, position / velocity targets, a hover-input iterate."""
import numpy as np

from .quad_config import QNX, QNU, QNY


def hover_input(cfg):
    return cfg.mass * cfg.g / (4.0 * cfg.max_thrust)


def random_quad_scenarios(B, cfg, seed=0, pos_err=1.5, tilt=0.3, aggressive=0.25):
    """x0: random position error, small random attitude (unit quaternion), velocity and body rates; reference: hover at the origin
    (a fraction `aggressive` of the instances gets a far target so that inputs saturate); iterate: x0 held, hover inputs."""
    rng = np.random.default_rng(seed)
    N = cfg.N
    x0 = np.zeros((B, QNX))
    x0[:, 0:3] = rng.uniform(-pos_err, pos_err, (B, 3))
    ax = rng.standard_normal((B, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, tilt, B)
    x0[:, 3] = np.cos(ang / 2); x0[:, 4:7] = ax * np.sin(ang / 2)[:, None]
    x0[:, 7:10] = rng.uniform(-1, 1, (B, 3)); x0[:, 10:13] = rng.uniform(-0.5, 0.5, (B, 3))
    far = rng.uniform(size=B) < aggressive
    x0[far, 0:3] *= 6.0
    yref = np.zeros((B, N, QNY)); yref[:, :, 3] = 1.0; yref[:, :, QNX:] = hover_input(cfg)
    yref_e = np.zeros((B, QNX)); yref_e[:, 3] = 1.0
    xbar = np.repeat(x0[:, None, :], N + 1, axis=1)
    ubar = np.full((B, N, QNU), hover_input(cfg)) + rng.uniform(-0.02, 0.02, (B, N, QNU))
    return dict(x0=x0, yref=yref, yref_e=yref_e, xbar=xbar, ubar=ubar)
