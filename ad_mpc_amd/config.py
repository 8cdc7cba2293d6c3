"""ctypes mirror of ``AdmpcConfig`` / ``AdmpcGp`` (include/admpc.h) plus the reference's shipped values.

Every default cites the reference line it comes from (paths relative to
``data_driven_mpc/ros_gp_mpc`` of HMCL-UNIST/AD_MPC).
"""
import ctypes as C

import numpy as np

NX, NU, NY = 7, 2, 9
MAX_N = 128
GP_MAX = 4
GP_MAX_POINTS = 32
GP_MAX_FEAT = 3


class AdmpcGp(C.Structure):
    _fields_ = [
        ("n_feat", C.c_int32),
        ("feat", C.c_int32 * GP_MAX_FEAT),
        ("out", C.c_int32),
        ("n_points", C.c_int32),
        ("sigma_f", C.c_double),
        ("inv_l2", C.c_double * GP_MAX_FEAT),
        ("ymean", C.c_double),
        ("Z", (C.c_double * GP_MAX_POINTS) * GP_MAX_FEAT),
        ("alpha", C.c_double * GP_MAX_POINTS),
    ]


class AdmpcConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("ipm_iter_max", C.c_int32),
        ("sqp_iters", C.c_int32),
        ("n_gp", C.c_int32),
        ("Ts", C.c_double),
        ("W", C.c_double * NY),
        ("We", C.c_double * NX),
        ("lbu", C.c_double * NU),
        ("ubu", C.c_double * NU),
        ("lbx_delta", C.c_double),
        ("ubx_delta", C.c_double),
        ("zl", C.c_double),
        ("zu", C.c_double),
        ("mass", C.c_double),
        ("L_F", C.c_double),
        ("L_R", C.c_double),
        ("Iz", C.c_double),
        ("Cf", C.c_double),
        ("Cr", C.c_double),
        ("ipm_mu0", C.c_double),
        ("ipm_thr0", C.c_double),
        ("ipm_tol_comp", C.c_double),
        ("ipm_tol_res", C.c_double),
        ("ipm_tol_step", C.c_double),
        ("ipm_try_unconstrained", C.c_double),
        ("ipm_warm_thr", C.c_double),
        ("ipm_warm_restart", C.c_double),
        ("ipm_fallback_iter", C.c_double),
        ("sqp_tol", C.c_double),
        ("gp", AdmpcGp * GP_MAX),
    ]

    def copy(self):
        other = AdmpcConfig()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(AdmpcConfig))
        return other


# --- vehicle constants: src/ad_mpc/ad_3d.py:47-71 (the 3.14195 "pi" is part of the model) -----------
VEH_MASS = 1500.0
VEH_F_MASS = 900.0
VEH_R_MASS = VEH_MASS - VEH_F_MASS
VEH_L = 2.7
VEH_L_F = VEH_L * (1 - VEH_F_MASS / VEH_MASS)
VEH_L_R = VEH_L * (1 - VEH_R_MASS / VEH_MASS)
VEH_IZ = VEH_L_F * VEH_L_R * (VEH_R_MASS + VEH_F_MASS)
VEH_CF = VEH_F_MASS * 0.5 * 9.81 * 0.165 * 180 / 3.14195
VEH_CR = VEH_R_MASS * 0.5 * 9.81 * 0.165 * 180 / 3.14195
BLEND_MAX = 110.0
BLEND_MIN = 100.0
STEERING_MIN, STEERING_MAX = -0.52, 0.52
STEERING_RATE_MIN, STEERING_RATE_MAX = -3.0, 3.0
ACC_MIN, ACC_MAX = -10.0, 5.0

# --- cost weights used by the node: src/ad_mpc/create_ros_ad_mpc.py:58-59 ---------------------------
Q_DIAG_ROS = (10.0, 10.0, 100.0, 0.0, 0.0, 0.0, 0.0)
R_DIAG_ROS = (1.0, 100.0)
# optimizer-level defaults (unused by the node): src/ad_mpc/ad_3d_optimizer.py:42-45
Q_DIAG_OPT = (10.0, 10.0, 50.0, 0.0, 0.0, 0.0, 1.0)
R_DIAG_OPT = (1.0, 100.0)
TERMINAL_SCALE = 1e-6      # ocp.cost.W_e = diag(q)*1e-6, src/ad_mpc/ad_3d_optimizer.py:151
SLACK_L1 = 10.0            # ocp.cost.zl = zu = 1e1, src/ad_mpc/ad_3d_optimizer.py:171-173

# --- interior point defaults (ours; the reference delegates to HPIPM mode BALANCE, iter_max 50,
#     c_generated_code/acados_solver_sim_car.c:688-692) ---------------------------------------------
IPM_ITER_MAX = 50
IPM_MU0 = 1.0
IPM_THR0 = 0.1
IPM_WARM_THR = 0.01
IPM_WARM_RESTART = 0.1
IPM_FALLBACK_ITER = 30.0
# Stopping levels of the interior point.  Default: the reference's own -- it leaves the QP tolerances unset (acados_models/sim_car_acados_ocp.json:
# qp_solver_tol_* null) and runs HPIPM in mode BALANCE (c_generated_code/acados_solver_sim_car.c:688): every residual norm, the
# complementarity products included, <= 1e-8, and no test on the step.  TIGHT: the levels of rounds 1-2 (complementarity 1e-10, residual 1e-9,
# last input step 1e-6), which take an instance with a nearly degenerate bound pair to within 1e-8 of the exact minimiser (the reference's
# levels leave such an instance up to ~1e-4 away from it: the error of a degenerate pair is sqrt(mu)); `tight_ipm(cfg)` sets them.
IPM_TOL_COMP = 1e-8
IPM_TOL_RES = 1e-8
IPM_TOL_STEP = 1e30
IPM_TIGHT = (1e-10, 1e-9, 1e-6)


def default_config(N=20, Ts=0.05, q=Q_DIAG_ROS, r=R_DIAG_ROS, terminal_scale=TERMINAL_SCALE, sqp_iters=1, sqp_tol=0.0):
    """The reference's shipped OCP for horizon ``N`` and sampling time ``Ts``."""
    if not (2 <= N <= MAX_N):
        raise ValueError("N must be in [2, %d]" % MAX_N)
    c = AdmpcConfig()
    c.N = int(N)
    c.ipm_iter_max = IPM_ITER_MAX
    c.sqp_iters = int(sqp_iters)
    c.sqp_tol = float(sqp_tol)
    c.n_gp = 0
    c.Ts = float(Ts)
    for i in range(NX):
        c.W[i] = float(q[i])
        c.We[i] = float(q[i]) * terminal_scale
    for j in range(NU):
        c.W[NX + j] = float(r[j])
    c.lbu[0], c.lbu[1] = ACC_MIN, STEERING_RATE_MIN
    c.ubu[0], c.ubu[1] = ACC_MAX, STEERING_RATE_MAX
    c.lbx_delta, c.ubx_delta = STEERING_MIN, STEERING_MAX
    c.zl = c.zu = SLACK_L1
    c.mass, c.L_F, c.L_R, c.Iz, c.Cf, c.Cr = VEH_MASS, VEH_L_F, VEH_L_R, VEH_IZ, VEH_CF, VEH_CR
    c.ipm_mu0, c.ipm_thr0 = IPM_MU0, IPM_THR0
    c.ipm_tol_comp, c.ipm_tol_res, c.ipm_tol_step = IPM_TOL_COMP, IPM_TOL_RES, IPM_TOL_STEP
    c.ipm_try_unconstrained = 1.0
    c.ipm_warm_thr = IPM_WARM_THR
    c.ipm_warm_restart = IPM_WARM_RESTART
    c.ipm_fallback_iter = IPM_FALLBACK_ITER
    return c


def set_gp(cfg, gps):
    """Install residual GPs.  ``gps``: iterable of dicts with keys feat (an index into [x;u] or a list of up to 3), out, Z (n or
    n x d training inputs), alpha, length_scale (scalar or one per feature), sigma_f, ymean -- squared-exponential GPs with a diagonal
    length-scale matrix, src/model_fitting/gp.py:81-138,446-471."""
    gps = list(gps)
    if len(gps) > GP_MAX:
        raise ValueError("at most %d GPs" % GP_MAX)
    cfg.n_gp = len(gps)
    for g, d in enumerate(gps):
        feats = [int(f) for f in np.atleast_1d(d["feat"]).reshape(-1)]
        nf = len(feats)
        al = np.asarray(d["alpha"], dtype=np.float64).reshape(-1)
        Z = np.asarray(d["Z"], dtype=np.float64).reshape(al.size, -1)
        ell = np.broadcast_to(np.asarray(d["length_scale"], dtype=np.float64).reshape(-1), (nf,)) if np.size(d["length_scale"]) in (1, nf) else None
        if not (1 <= nf <= GP_MAX_FEAT) or Z.shape[1] != nf or ell is None or al.size > GP_MAX_POINTS:
            raise ValueError("bad GP size")
        s = cfg.gp[g]
        s.n_feat, s.out, s.n_points = nf, int(d["out"]), int(al.size)
        s.sigma_f = float(d.get("sigma_f", 1.0))
        s.ymean = float(d.get("ymean", 0.0))
        for k in range(GP_MAX_FEAT):
            s.feat[k] = feats[k] if k < nf else 0
            s.inv_l2[k] = 1.0 / float(ell[k]) ** 2 if k < nf else 0.0
            for i in range(GP_MAX_POINTS):
                s.Z[k][i] = float(Z[i, k]) if (k < nf and i < al.size) else 0.0
        for i in range(GP_MAX_POINTS):
            s.alpha[i] = float(al[i]) if i < al.size else 0.0
    return cfg


def tight_ipm(cfg):
    """The interior point's stopping levels of rounds 1-2 (in place; returns cfg): every instance to within 1e-8 of the exact minimiser."""
    cfg.ipm_tol_comp, cfg.ipm_tol_res, cfg.ipm_tol_step = IPM_TIGHT
    return cfg


def tight_config(*a, **kw):
    """default_config with the tight stopping levels (tests that compare with exact minimisers, iteration-count tables of rounds 1-2)."""
    return tight_ipm(default_config(*a, **kw))
