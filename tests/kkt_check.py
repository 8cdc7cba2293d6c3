"""Independent (numpy) KKT residual evaluation for the QP of one RTI step.

Used by the tests to prove that a returned step is THE minimiser of the strictly convex QP without
trusting the interior-point code that produced it: multipliers of the dynamics are reconstructed
by the adjoint recursion, then stationarity, primal/dual feasibility and complementarity are
evaluated explicitly.
"""
import numpy as np

NX, NU = 7, 2


def kkt_residuals(cfg, x0, yref, yref_e, xbar, ubar, dbg):
    N = cfg.N
    Ts = cfg.Ts
    W = np.array(cfg.W[:]); We = np.array(cfg.We[:])
    Q = Ts * W[:NX]; R = Ts * W[NX:]
    rho_l, rho_u = Ts * cfg.zl, Ts * cfg.zu
    A, B, b, du, dx = dbg["A"], dbg["B"], dbg["b"], dbg["du"], dbg["dx"]
    lam_u, lam_d, sl, su = dbg["lam_u"], dbg["lam_d"], dbg["sl"], dbg["su"]
    q = Q * (xbar[:N] - yref[:, :NX]); r = R * (ubar - yref[:, NX:])
    qN = We * (xbar[N] - yref_e)
    res = {}
    res["x0"] = np.abs(dx[0] - (x0 - xbar[0])).max()
    dyn = np.array([A[k] @ dx[k] + B[k] @ du[k] + b[k] - dx[k + 1] for k in range(N)])
    res["dyn"] = np.abs(dyn).max()
    pi = np.zeros((N, NX))
    pi[N - 1] = We * dx[N] + qN
    for k in range(N - 1, 0, -1):
        g = Q * dx[k] + q[k]
        g[6] += -lam_d[k, 0] + lam_d[k, 1]
        pi[k - 1] = A[k].T @ pi[k] + g
    stat_u = np.array([R * du[k] + r[k] + B[k].T @ pi[k] - lam_u[k, :, 0] + lam_u[k, :, 1] for k in range(N)])
    res["stat_u"] = np.abs(stat_u).max()
    res["stat_s"] = max(np.abs(rho_l - lam_u[:, :, 0] - lam_u[:, :, 2]).max(), np.abs(rho_u - lam_u[:, :, 1] - lam_u[:, :, 3]).max())
    lbu = np.array(cfg.lbu[:]); ubu = np.array(cfg.ubu[:])
    t0 = du + sl - (lbu - ubar); t1 = -du + su + (ubu - ubar)
    t5 = dx[1:N, 6] - (cfg.lbx_delta - xbar[1:N, 6]); t6 = (cfg.ubx_delta - xbar[1:N, 6]) - dx[1:N, 6]
    res["prim"] = max(0.0, -min(t0.min(), t1.min(), sl.min(), su.min(), t5.min(), t6.min()))
    res["dual"] = max(0.0, -min(lam_u.min(), lam_d[1:N].min()))
    res["comp"] = max(np.abs(lam_u[:, :, 0] * t0).max(), np.abs(lam_u[:, :, 1] * t1).max(), np.abs(lam_u[:, :, 2] * sl).max(),
                      np.abs(lam_u[:, :, 3] * su).max(), np.abs(lam_d[1:N, 0] * t5).max(), np.abs(lam_d[1:N, 1] * t6).max())
    return res
