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


def kkt_residuals_from_multipliers(cfg, x0, yref, yref_e, xbar, ubar, A, B, phi, x_new, u_new, pi, ineq):
    """KKT residuals of the QP of one RTI step evaluated with the multipliers the solver itself returns
    (admpc_solve_batch_ex: pi [N+1,7], ineq [N,20], record order of include/admpc.h) -- nothing is reconstructed here, so the
    stationarity rows test the returned multipliers, not an adjoint recursion of this file.  Linearisation (A, B, phi) at the
    iterate the step started from (xbar, ubar)."""
    N = cfg.N
    Ts = cfg.Ts
    W = np.array(cfg.W[:]); We = np.array(cfg.We[:])
    Q = Ts * W[:NX]; R = Ts * W[NX:]
    rho_l, rho_u = Ts * cfg.zl, Ts * cfg.zu
    dx = x_new - xbar; du = u_new - ubar
    b = phi - xbar[1:]
    t, lam = ineq[:, :10], ineq[:, 10:]
    q = Q * (xbar[:N] - yref[:, :NX]); r = R * (ubar - yref[:, NX:])
    qN = We * (xbar[N] - yref_e)
    res = {}
    res["x0"] = np.abs(dx[0] - (x0 - xbar[0])).max()
    res["dyn"] = np.abs(np.array([A[k] @ dx[k] + B[k] @ du[k] + b[k] - dx[k + 1] for k in range(N)])).max()
    sx = [np.abs(We * dx[N] + qN - pi[N - 1]).max()]
    for k in range(1, N):
        g = Q * dx[k] + q[k] + A[k].T @ pi[k] - pi[k - 1]
        g[6] += -lam[k, 4] + lam[k, 5]
        sx.append(np.abs(g).max())
    res["stat_x"] = max(sx)
    res["stat_x0"] = np.abs(Q * dx[0] + q[0] + A[0].T @ pi[0] - pi[N]).max()        # pi[N]: multiplier of the initial-state equality
    su_ = []
    for k in range(N):
        g = R * du[k] + r[k] + B[k].T @ pi[k]
        g[0] += -lam[k, 0] + lam[k, 1]; g[1] += -lam[k, 2] + lam[k, 3]
        su_.append(np.abs(g).max())
    res["stat_u"] = max(su_)
    res["stat_s"] = max(np.abs(rho_l - lam[:, 0] - lam[:, 6]).max(), np.abs(rho_u - lam[:, 1] - lam[:, 7]).max(),
                        np.abs(rho_l - lam[:, 2] - lam[:, 8]).max(), np.abs(rho_u - lam[:, 3] - lam[:, 9]).max())
    lbu = np.array(cfg.lbu[:]); ubu = np.array(cfg.ubu[:])
    # slacks as the constraints define them, from the returned primal point and slack variables (t[6..9] are sl0, su0, sl1, su1)
    sl = np.stack([t[:, 6], t[:, 8]], axis=1); su = np.stack([t[:, 7], t[:, 9]], axis=1)
    tl = du + sl - (lbu - ubar); tu = -du + su + (ubu - ubar)
    t5 = dx[1:N, 6] - (cfg.lbx_delta - xbar[1:N, 6]); t6 = (cfg.ubx_delta - xbar[1:N, 6]) - dx[1:N, 6]
    res["slack_consistency"] = max(np.abs(tl[:, 0] - t[:, 0]).max(), np.abs(tu[:, 0] - t[:, 1]).max(), np.abs(tl[:, 1] - t[:, 2]).max(), np.abs(tu[:, 1] - t[:, 3]).max(),
                                   np.abs(t5 - t[1:N, 4]).max(), np.abs(t6 - t[1:N, 5]).max())
    res["prim"] = max(0.0, -min(tl.min(), tu.min(), sl.min(), su.min(), t5.min(), t6.min()))
    res["dual"] = max(0.0, -lam.min())
    res["comp"] = np.abs(lam * t)[1:].max(initial=0.0) if N > 1 else 0.0
    res["comp"] = max(res["comp"], np.abs(lam[0, [0, 1, 2, 3, 6, 7, 8, 9]] * t[0, [0, 1, 2, 3, 6, 7, 8, 9]]).max())
    return res


def nlp_residuals_numpy(cfg, x0, yref, yref_e, xbar, ubar, A, B, phi, pi, ineq):
    """acados' four SQP stopping residuals (res_stat, res_eq, res_ineq, res_comp: inf-norms of the rows of the NLP's KKT system) at the
    iterate (xbar, ubar), written out independently of the C restatement: A, B, phi = linearisation AT that iterate, pi [N+1,7] and ineq
    [N,20] in the record order of include/admpc.h.  The slack variables are read from the slacks of their own bounds (t[6..9])."""
    N = cfg.N
    Ts = cfg.Ts
    W = np.array(cfg.W[:]); We = np.array(cfg.We[:])
    Q = Ts * W[:NX]; R = Ts * W[NX:]
    rho_l, rho_u = Ts * cfg.zl, Ts * cfg.zu
    lbu = np.array(cfg.lbu[:]); ubu = np.array(cfg.ubu[:])
    t, lam = ineq[:, :10], ineq[:, 10:]
    stat = [np.abs(We * (xbar[N] - yref_e) - pi[N - 1]).max()]
    ineq_r, comp = [], []
    for k in range(N):
        g = Q * (xbar[k] - yref[k, :NX]) + A[k].T @ pi[k] - (pi[k - 1] if k >= 1 else pi[N])
        if k >= 1:
            g[6] += -lam[k, 4] + lam[k, 5]
        gu = R * (ubar[k] - yref[k, NX:]) + B[k].T @ pi[k]
        gu[0] += -lam[k, 0] + lam[k, 1]; gu[1] += -lam[k, 2] + lam[k, 3]
        gs = [rho_l - lam[k, 0] - lam[k, 6], rho_u - lam[k, 1] - lam[k, 7], rho_l - lam[k, 2] - lam[k, 8], rho_u - lam[k, 3] - lam[k, 9]]
        stat.append(max(np.abs(g).max(), np.abs(gu).max(), np.abs(gs).max()))
        for j in range(NU):
            ineq_r += [ubar[k, j] + t[k, 6 + 2 * j] - lbu[j] - t[k, 2 * j], ubu[j] - ubar[k, j] + t[k, 7 + 2 * j] - t[k, 2 * j + 1]]
        idx = [0, 1, 2, 3, 6, 7, 8, 9]
        if k >= 1:
            ineq_r += [xbar[k, 6] - cfg.lbx_delta - t[k, 4], cfg.ubx_delta - xbar[k, 6] - t[k, 5]]
            idx += [4, 5]
        comp.append(np.abs(lam[k, idx] * t[k, idx]).max())
    eq = max(np.abs(phi - xbar[1:]).max(), np.abs(x0 - xbar[0]).max())
    return np.array([max(stat), eq, np.abs(ineq_r).max(), max(comp)])
