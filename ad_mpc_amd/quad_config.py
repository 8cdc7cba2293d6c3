"""ctypes mirror of include/admpc_quad.h (AdmpcQuadConfig) and the shipped quadrotor problem
(acados_models/my_quad_acados_ocp.json, src/quad_mpc/quad_3d.py:40-74)."""
import ctypes as C
import math

import numpy as np

from .config import AdmpcGp, GP_MAX_FEAT, GP_MAX_POINTS

QNX, QNU, QNY, QUAD_MAX_N, QUAD_GP_MAX = 13, 4, 17, 24, 3


class AdmpcQuadConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("ipm_iter_max", C.c_int32), ("Ts", C.c_double),
        ("W", C.c_double * QNY), ("We", C.c_double * QNX),
        ("lbu", C.c_double * QNU), ("ubu", C.c_double * QNU),
        ("mass", C.c_double), ("J", C.c_double * 3), ("max_thrust", C.c_double),
        ("x_f", C.c_double * 4), ("y_f", C.c_double * 4), ("z_l_tau", C.c_double * 4), ("g", C.c_double), ("rdrv", C.c_double * 3),
        ("ipm_mu0", C.c_double), ("ipm_thr0", C.c_double), ("ipm_tol_comp", C.c_double), ("ipm_tol_res", C.c_double),
        ("sqp_tol", C.c_double),
        ("n_gp", C.c_int32), ("sqp_iters", C.c_int32), ("gp", AdmpcGp * QUAD_GP_MAX),
    ]

    def copy(self):
        c = AdmpcQuadConfig()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(self))
        return c


def default_quad_config(N=10, t_horizon=1.0):
    """my_quad_acados_ocp.json: N = 10, tf = 1 s, W = diag(10,10,10, 0,.1,.1,.1, .05 x 6, .1 x 4), W_e = 0, 0 <= u <= 1;
    vehicle of quad_3d.py ('x' configuration)."""
    if not (2 <= N <= QUAD_MAX_N):
        raise ValueError("N must be in [2, %d]" % QUAD_MAX_N)
    c = AdmpcQuadConfig()
    c.N, c.ipm_iter_max, c.Ts = int(N), 50, float(t_horizon) / N
    for i, v in enumerate([10.0] * 3 + [0.0] + [0.1] * 3 + [0.05] * 6 + [0.1] * 4):
        c.W[i] = v
    for i in range(QNX):
        c.We[i] = 0.0
    for i in range(QNU):
        c.lbu[i], c.ubu[i] = 0.0, 1.0
    c.mass, c.max_thrust, c.g = 1.0, 20.0, 9.81
    c.J[0], c.J[1], c.J[2] = 0.03, 0.03, 0.06
    h = math.cos(math.pi / 4) * (0.47 / 2)
    for i, (xf, yf, zt) in enumerate(zip((h, -h, -h, h), (-h, -h, h, h), (-0.013, 0.013, -0.013, 0.013))):
        c.x_f[i], c.y_f[i], c.z_l_tau[i] = xf, yf, zt
    c.ipm_mu0, c.ipm_thr0 = 1.0, 0.1
    c.sqp_iters, c.sqp_tol = 1, 0.0                 # SQP_RTI (the shipped solver_type); "SQP": sqp_iters 100, sqp_tol 1e-6 (my_quad_acados_ocp.json:2075-2080)
    c.ipm_tol_comp, c.ipm_tol_res = 1e-8, 1e-8      # the reference's levels: HPIPM mode BALANCE (acados_models/my_quad_acados_ocp.json leaves qp_solver_tol_* unset); tight_quad_ipm: 1e-10 / 1e-9
    return c


def set_quad_gp(cfg, gps):
    """Residual GPs of the quadrotor (quad_3d_optimizer.py:289-327).  ``gps``: dicts as for ``config.set_gp``; ``feat`` indexes
    z = [x with the velocity in the body frame (13); u (4)], ``out`` in {7, 8, 9} is the body-frame acceleration component."""
    from .config import AdmpcConfig, set_gp
    gps = list(gps)
    if len(gps) > QUAD_GP_MAX:
        raise ValueError("at most %d GPs" % QUAD_GP_MAX)
    tmp = AdmpcConfig()
    set_gp(tmp, gps)                              # fills AdmpcGp entries (same struct)
    for g, d in enumerate(gps):
        feats = [int(f) for f in np.atleast_1d(d["feat"]).reshape(-1)]
        if not (7 <= int(d["out"]) <= 9) or any(not (7 <= f < QNX + QNU) for f in feats):
            raise ValueError("quadrotor GP: out must be in {7, 8, 9}, features in [7, 17) (body-frame velocity, body rates, inputs)")
        C.memmove(C.byref(cfg.gp[g]), C.byref(tmp.gp[g]), C.sizeof(AdmpcGp))
    cfg.n_gp = len(gps)
    return cfg


def tight_quad_ipm(cfg):
    """The stop levels of earlier rounds (complementarity 1e-10, residual 1e-9), in place; returns cfg."""
    cfg.ipm_tol_comp, cfg.ipm_tol_res = 1e-10, 1e-9
    return cfg


def tight_quad_config(*a, **kw):
    return tight_quad_ipm(default_quad_config(*a, **kw))
