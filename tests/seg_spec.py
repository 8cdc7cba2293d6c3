"""Executable specification (numpy, TEST INFRASTRUCTURE) of the segmented condensed interior point that
`ad_mpc_amd/csrc/admpc_seg.hip` runs for horizons N = S x 20 (the reference's shipped N = 40, BASELINE configs[4]'s N = 80).

It restates, in plain dense linear algebra, the block elimination the kernel performs with S cooperating waves, so that the
MATHEMATICS of the kernel (segment condensing, bordered LDL', the 7 x 7 interface recursion, start rules, stopping test) is
checked on the CPU against the stage-wise Riccati oracle (`oracle/admpc_oracle.c:ipm_solve`) -- same Newton steps, another
elimination order -- before and independently of any device run.  Nothing in the product imports this file.

The QP (reference: acados FULL_CONDENSING_HPIPM, acados_solver_sim_car.c:145,688-692) is split into S segments of
Ns stages.  Segment s owns the inputs U_s of its stages, the states of its stages as functions of (z_s, U_s) with z_s = dx at the
segment's first stage (z_0 = x0 - xbar_0 is data), the delta boxes of its stages and its tracking cost; consecutive segments are
coupled by  z_{s+1} = Bbar_s U_s + Abar_s z_s + c_s  with multiplier nu_{s+1}.  Per interior-point iteration every segment
factorises its own 40 x 40 Newton matrix M_s = L D L' with the border rows C_s = [Qzu_s ; Bbar_s] riding along
(L_b = C L^-T D^-1), which yields the Schur blocks  Pzz = Qzz - Qzu M^-1 Quz,  Pzb = Qzu M^-1 Bbar',  Pbb = Bbar M^-1 Bbar';
a backward recursion over the S - 1 interfaces (7 x 7 blocks, Gaussian elimination with partial pivoting on
Lambda = I + Pbb Pi) couples them, and each segment back-substitutes its own inputs.
"""
import numpy as np

NX, NU = 7, 2
IPM_FLOOR = 1e-40
BLOCKED_STEP = 0.05
MU_FLOOR = 1e-3
FLOOR_CAP = 1e3


class SegQP:
    """Stage data of one RTI step (what oracle build_qp holds) + the per-segment condensed form."""

    def __init__(self, cfg, A, B, b, x0, yref, yref_e, xbar, ubar, Ns=20):
        N = cfg.N
        assert N % Ns == 0
        self.N, self.Ns, self.S = N, Ns, N // Ns
        Ts = cfg.Ts
        self.h = Ts
        W = np.array(cfg.W[:]); We = np.array(cfg.We[:])
        self.Qd = Ts * W[:NX]; self.Rd = Ts * W[NX:]; self.Qe = We.copy()
        self.A, self.B, self.b = A, B, b
        self.q = np.empty((N + 1, NX)); self.r = np.empty((N, NU))
        self.q[:N] = self.Qd * (xbar[:N] - yref[:, :NX]); self.q[N] = self.Qe * (xbar[N] - yref_e)
        self.r[:] = self.Rd * (ubar - yref[:, NX:])
        lbu = np.array(cfg.lbu[:]); ubu = np.array(cfg.ubu[:])
        self.dlu = lbu - ubar; self.duu = ubu - ubar
        self.dld = cfg.lbx_delta - xbar[:N, 6]; self.dud = cfg.ubx_delta - xbar[:N, 6]
        self.dx0 = x0 - xbar[0]
        self.rho_l = Ts * cfg.zl; self.rho_u = Ts * cfg.zu
        self.segs = [self._condense(s) for s in range(self.S)]

    def _condense(self, s):
        Ns, S, N = self.Ns, self.S, self.N
        n = NU * Ns
        k0 = s * Ns
        first, last = s == 0, s == S - 1
        G = np.zeros((NX, n)); P = np.zeros((NX, NX)) if first else np.eye(NX)
        xh = self.dx0.copy() if first else np.zeros(NX)
        Huu = np.zeros((n, n)); Hzu = np.zeros((NX, n)); Hzz = np.zeros((NX, NX))
        gu = self.r[k0:k0 + Ns].reshape(-1).copy(); gz = np.zeros(NX)
        xh6 = np.zeros(Ns)                       # free response of delta at the segment's stages
        rx0 = 0.0                                # max |Qd dx + q| over the segment's cost stages at (U = 0, z = cold chain): see cold_rstat
        stages = []                              # (Gamma_k, Phi_k, xhat_k, weight) of the cost stages, for the cold-start residual
        for kk in range(Ns + 1):
            k = k0 + kk
            cost_here = (kk >= 1 or not first) and (kk < Ns or last)
            if cost_here:
                w = self.Qd if k < N else self.Qe
                Huu += G.T @ (w[:, None] * G); Hzu += P.T @ (w[:, None] * G); Hzz += P.T @ (w[:, None] * P)
                gk = w * xh + self.q[k]
                gu += G.T @ gk; gz += P.T @ gk
                stages.append((k, P.copy(), xh.copy(), w))
            if kk < Ns:
                xh6[kk] = xh[6]
                Gn = self.A[k] @ G
                Gn[:, NU * kk:NU * kk + NU] = self.B[k]
                G, P, xh = Gn, self.A[k] @ P, self.A[k] @ xh + self.b[k]
        return dict(s=s, k0=k0, first=first, last=last, Huu=Huu, Hzu=Hzu, Hzz=Hzz, gu=gu, gz=gz, xh6=xh6,
                    Bbar=G, Abar=P, c=xh, stages=stages)


def _ldl(M):
    """square-root-free M = L D L' without pivoting (what the kernel's right-looking factorisation computes)."""
    n = M.shape[0]
    a = M.copy(); L = np.eye(n); d = np.empty(n)
    for j in range(n):
        d[j] = a[j, j]
        L[j + 1:, j] = a[j + 1:, j] / d[j]
        a[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], a[j + 1:, j])
    return L, d


class SegState:
    pass


def seg_ipm(cfg, qp, trace=None, dbg=None):
    """Returns (du [N,2], dx [N+1,7], iters); iters < 0 on a non-finite iterate.  Mirrors oracle ipm_solve line by line in its
    control flow (trial, warm start, restart, fallback, safeguards, stopping test with the tracked stationarity residual)."""
    S, Ns, h = qp.S, qp.Ns, qp.h
    n = NU * Ns
    thr, mu0 = cfg.ipm_thr0, cfg.ipm_mu0
    n_ineq = 8 * qp.N + 2 * (qp.N - 1)
    segs = qp.segs
    odd = np.arange(n) % 2 == 1
    kidx = np.arange(n) // 2                       # local stage of input i

    def dact(sg):                                  # local stages that carry a delta box (global stage >= 1; none at stage N)
        m = np.ones(Ns, dtype=bool)
        if sg["first"]:
            m[0] = False
        return m

    def dbounds(sg):
        k0 = sg["k0"]
        return qp.dld[k0:k0 + Ns], qp.dud[k0:k0 + Ns]

    def dx6_of(sg, U, z):                          # delta step at the segment's stages
        u1 = U[1::2]
        pre = np.concatenate([[0.0], np.cumsum(u1)[:-1]])
        return (0.0 if sg["first"] else z[6]) + sg["xh6"] + h * pre

    def chain_z(Us):                               # interface states of given inputs (z_0 is data inside segment 0)
        zs = [np.zeros(NX)]
        for s in range(S - 1):
            sg = segs[s]
            zs.append(sg["Bbar"] @ Us[s] + (sg["Abar"] @ zs[s] if s > 0 else 0.0) + sg["c"])
        return zs

    st = SegState()

    def cold_start():
        st.U = [np.zeros(n) for _ in range(S)]
        st.z = chain_z(st.U)
        st.sl = [np.full(n, thr) for _ in range(S)]; st.su = [np.full(n, thr) for _ in range(S)]
        st.t = []; st.lam = []; st.td = []; st.lamd = []
        for s, sg in enumerate(segs):
            k0 = sg["k0"]
            dl = qp.dlu[k0:k0 + Ns].reshape(-1); du_ = qp.duu[k0:k0 + Ns].reshape(-1)
            r0 = np.stack([thr - dl, thr + du_, np.full(n, thr), np.full(n, thr)], 1)
            t = np.maximum(r0, thr); st.t.append(t); st.lam.append(mu0 / t)
            dld, dud = dbounds(sg)
            d6 = dx6_of(sg, st.U[s], st.z[s])
            r0 = np.stack([d6 - dld, dud - d6], 1)
            td = np.where(dact(sg)[:, None], np.maximum(r0, thr), 1.0)
            st.td.append(td); st.lamd.append(np.where(dact(sg)[:, None], mu0 / td, 0.0))

    def local_grad(s):
        """gradient of the segment's own cost + inequality multipliers at the current point, WITHOUT the interface multipliers
        (the primal Newton step does not depend on them: oracle comment above riccati_solve / rowqp_core.h)."""
        sg = segs[s]; U, z = st.U[s], st.z[s]
        lam, lamd = st.lam[s], st.lamd[s]
        da = dact(sg)
        dl = np.where(da, lamd[:, 1] - lamd[:, 0], 0.0)
        suf = np.cumsum(dl[::-1])[::-1] - dl                         # sum over stages > k
        gU = sg["Huu"] @ U + qp.Rd[np.arange(n) % 2] * U + sg["gu"] - lam[:, 0] + lam[:, 1] + np.where(odd, h * suf[kidx], 0.0)
        gz = None
        if not sg["first"]:
            gU = gU + sg["Hzu"].T @ z
            gz = sg["Hzu"] @ U + sg["Hzz"] @ z + sg["gz"]
            gz[6] += dl.sum()
        return gU, gz

    def start_rstat(warm_pi):
        """max-norm of the stationarity rows (ru, rx of the oracle) at the start point.  Cold start: dynamics multipliers zero.
        Warm start: the multipliers are the trial's, which make the cost part stationary: what is left is the inequality part."""
        m = 0.0
        for s, sg in enumerate(segs):
            lam, lamd = st.lam[s], st.lamd[s]
            da = dact(sg)
            if warm_pi:
                ru = -lam[:, 0] + lam[:, 1]
                m = max(m, np.abs(ru).max(), np.abs(np.where(da, lamd[:, 1] - lamd[:, 0], 0.0)).max())
            else:
                k0 = sg["k0"]
                ru = qp.r[k0:k0 + Ns].reshape(-1) - lam[:, 0] + lam[:, 1]
                m = max(m, np.abs(ru).max())
                for (k, P, xh, w) in sg["stages"]:
                    dxk = xh + (P @ st.z[s] if not sg["first"] else 0.0)
                    rx = w * dxk + qp.q[k]
                    if k < qp.N and k >= 1:
                        kk = k - k0
                        rx[6] += lamd[kk, 1] - lamd[kk, 0]
                    m = max(m, np.abs(rx).max())
        return m

    def newton(Rt_list, G56_list, yU_list, yz_list, fac=None):
        """Solve the coupled Newton system; returns (dU list, dz list, factors for reuse)."""
        if fac is None:
            fac = []
            for s, sg in enumerate(segs):
                G56 = G56_list[s]
                suf = np.cumsum(G56[::-1])[::-1] - G56             # sum over stages m > k
                M = sg["Huu"] + np.diag(Rt_list[s])
                Sm = h * h * suf[np.maximum.outer(kidx, kidx)]
                M = M + np.where(np.outer(odd, odd), Sm, 0.0)
                rows = []
                if not sg["first"]:
                    Qzu = sg["Hzu"].copy()
                    Qzu[6] += np.where(odd, h * suf[kidx], 0.0)
                    Qzz = sg["Hzz"].copy(); Qzz[6, 6] += G56.sum()
                    rows.append(Qzu)
                else:
                    Qzu = None; Qzz = None
                if not sg["last"]:
                    rows.append(sg["Bbar"])
                L, d = _ldl(M)
                C = np.vstack(rows) if rows else np.zeros((0, n))
                Lb = np.linalg.solve(L, C.T).T / d                   # C L^-T D^-1
                Sc = Lb @ (d[:, None] * Lb.T)                        # C M^-1 C'
                nz = 0 if sg["first"] else NX
                f = dict(L=L, d=d, Lb=Lb, nz=nz, Sc=Sc, M=M, C=C)
                if not sg["first"]:
                    f["Pzz"] = Qzz - Sc[:NX, :NX]
                if not sg["last"]:
                    f["Pbb"] = Sc[nz:, nz:]
                    f["Ahat"] = (sg["Abar"] - Sc[:NX, nz:].T) if not sg["first"] else None
                fac.append(f)
            # interface matrices, backward
            Pi = [None] * S; Lam = [None] * S
            for s in range(S - 1, 0, -1):
                f = fac[s]
                if segs[s]["last"]:
                    Pi[s] = f["Pzz"]
                else:
                    Lam[s] = np.eye(NX) + f["Pbb"] @ Pi[s + 1]
                    Pi[s] = f["Pzz"] + f["Ahat"].T @ Pi[s + 1] @ np.linalg.solve(Lam[s], f["Ahat"])
            if S > 1:
                Lam[0] = np.eye(NX) + fac[0]["Pbb"] @ Pi[1]
            fac = dict(seg=fac, Pi=Pi, Lam=Lam)
        F = fac["seg"]; Pi, Lam = fac["Pi"], fac["Lam"]
        zeta = []; zb = []
        for s in range(S):
            f = F[s]
            ze = np.linalg.solve(f["L"], yU_list[s])
            yb = np.concatenate([yz_list[s] if f["nz"] else np.zeros(0), np.zeros(f["Lb"].shape[0] - f["nz"])])
            zeta.append(ze); zb.append(yb - f["Lb"] @ ze)
        # backward sweep of the interface right-hand sides: nu_s = eta_s - Pi_s dz_s
        eta = [None] * (S + 1)
        for s in range(S - 1, 0, -1):
            f = F[s]; yhat = zb[s][:NX]
            if segs[s]["last"]:
                eta[s] = yhat
            else:
                dhat = -zb[s][NX:]
                eta[s] = yhat + f["Ahat"].T @ (eta[s + 1] - Pi[s + 1] @ np.linalg.solve(Lam[s], dhat + f["Pbb"] @ eta[s + 1]))
        dz = [np.zeros(NX)] * S; nu = [np.zeros(NX)] * (S + 1)
        for s in range(S - 1):
            f = F[s]
            dhat = -zb[s][f["nz"]:]
            rhs = dhat + f["Pbb"] @ eta[s + 1] + (f["Ahat"] @ dz[s] if s > 0 else 0.0)
            dz[s + 1] = np.linalg.solve(Lam[s], rhs)
            nu[s + 1] = eta[s + 1] - Pi[s + 1] @ dz[s + 1]
        dU = []
        for s in range(S):
            f = F[s]
            w = np.concatenate([dz[s] if f["nz"] else np.zeros(0), -nu[s + 1] if not segs[s]["last"] else np.zeros(0)])
            x = zeta[s] / f["d"] - f["Lb"].T @ w
            dU.append(np.linalg.solve(f["L"].T, x))
        if dbg is not None:
            dbg.append(dict(fac=fac, zb=zb, dz=dz, nu=nu, dU=dU, yU=yU_list, yz=yz_list, eta=eta))
        return dU, dz, fac

    cold_start()
    warmed = False; cons = False
    if cfg.ipm_try_unconstrained != 0:
        yU = []; yz = []
        for s in range(S):
            sg = segs[s]
            # gradient of the cost alone (no inequality multipliers) at the cold start
            U, z = st.U[s], st.z[s]
            gU = sg["gu"].copy(); gz = None
            if not sg["first"]:
                gU = gU + sg["Hzu"].T @ z
                gz = sg["Hzz"] @ z + sg["gz"]
            yU.append(-gU); yz.append(-gz if gz is not None else None)
        dU, dz, _ = newton([qp.Rd[np.arange(n) % 2]] * S, [np.zeros(Ns)] * S, yU, yz)
        ok = True
        for s, sg in enumerate(segs):
            k0 = sg["k0"]
            dl = qp.dlu[k0:k0 + Ns].reshape(-1); du_ = qp.duu[k0:k0 + Ns].reshape(-1)
            ok = ok and bool(np.all((dU[s] >= dl) & (dU[s] <= du_)))
            d6 = dx6_of(sg, dU[s], st.z[s] + dz[s])
            dld, dud = dbounds(sg)
            ok = ok and bool(np.all(~dact(sg) | ((d6 >= dld) & (d6 <= dud))))
        if ok:
            st.U = dU
            return _finish(qp, st.U), 0
        if cfg.ipm_warm_thr > 0:
            warmed = True
            thw = cfg.ipm_warm_thr
            for s, sg in enumerate(segs):
                k0 = sg["k0"]
                dl = qp.dlu[k0:k0 + Ns].reshape(-1); du_ = qp.duu[k0:k0 + Ns].reshape(-1)
                v = dU[s]
                st.U[s] = v.copy(); st.z[s] = st.z[s] + dz[s]
                st.sl[s] = np.maximum(dl - v, 0.0) + thw; st.su[s] = np.maximum(v - du_, 0.0) + thw
                r0 = np.stack([v + st.sl[s] - dl, -v + st.su[s] + du_, st.sl[s], st.su[s]], 1)
                st.t[s] = np.maximum(r0, thw); st.lam[s] = mu0 / st.t[s]
                d6 = dx6_of(sg, st.U[s], st.z[s]); dld, dud = dbounds(sg)
                r0 = np.stack([d6 - dld, dud - d6], 1)
                st.td[s] = np.where(dact(sg)[:, None], np.maximum(r0, thw), 1.0)
                st.lamd[s] = np.where(dact(sg)[:, None], mu0 / st.td[s], 0.0)

    rmax_prev = 0.0; step = 1e300; alpha_prev = 1.0; rstat = -1.0
    it = 0
    itmax = cfg.ipm_iter_max
    fbit = int(cfg.ipm_fallback_iter)
    while it < itmax + (fbit if cons else 0):
        restart_here = True
        while restart_here:
            restart_here = False
            musum = 0.0; cmax = 0.0; rineq = 0.0
            loc = []
            for s, sg in enumerate(segs):
                k0 = sg["k0"]; da = dact(sg)
                t, lam, td, lamd = st.t[s], st.lam[s], st.td[s], st.lamd[s]
                U = st.U[s]
                dl = qp.dlu[k0:k0 + Ns].reshape(-1); du_ = qp.duu[k0:k0 + Ns].reshape(-1)
                rc = t * lam; rcd = np.where(da[:, None], td * lamd, 0.0)
                musum += rc.sum() + rcd.sum(); cmax = max(cmax, rc.max(), rcd.max())
                rd = np.stack([U + st.sl[s] - dl - t[:, 0], -U + st.su[s] + du_ - t[:, 1], st.sl[s] - t[:, 2], st.su[s] - t[:, 3]], 1)
                rsl = qp.rho_l - lam[:, 0] - lam[:, 2]; rsu = qp.rho_u - lam[:, 1] - lam[:, 3]
                d6 = dx6_of(sg, U, st.z[s]); dld, dud = dbounds(sg)
                rdd = np.where(da[:, None], np.stack([d6 - dld - td[:, 0], dud - d6 - td[:, 1]], 1), 0.0)
                rineq = max(rineq, np.abs(rsl).max(), np.abs(rsu).max(), np.abs(rd[:, :2]).max(), np.abs(rdd).max())
                loc.append(dict(rc=rc, rcd=rcd, rd=rd, rsl=rsl, rsu=rsu, rdd=rdd))
            mu = musum / n_ineq
            if not np.isfinite(mu) or not np.isfinite(rineq):
                return None, -1
            if rstat < 0:
                rstat = start_rstat(warmed)
            rmax = max(rineq, rstat)
            if cmax <= cfg.ipm_tol_comp and step <= cfg.ipm_tol_step and \
                    (rmax <= cfg.ipm_tol_res or (it > 0 and rmax > 0.1 * rmax_prev and rmax <= FLOOR_CAP * cfg.ipm_tol_res)):
                return _finish(qp, st.U), it
            rmax_prev = rmax
            if not cons and fbit > 0 and it >= fbit:
                cons = True; warmed = False
                cold_start()
                alpha_prev = 1.0; step = 1e300; rmax_prev = 0.0; rstat = -1.0
                restart_here = True
        # barrier quantities
        Rt = []; G56l = []; aux = []
        for s, sg in enumerate(segs):
            t, lam, td, lamd = st.t[s], st.lam[s], st.td[s], st.lamd[s]
            G = lam / t
            Rt.append(qp.Rd[np.arange(n) % 2] + G[:, 0] * G[:, 2] / (G[:, 0] + G[:, 2]) + G[:, 1] * G[:, 3] / (G[:, 1] + G[:, 3]))
            da = dact(sg)
            G5 = np.where(da, lamd[:, 0] / td[:, 0], 0.0); G6 = np.where(da, lamd[:, 1] / td[:, 1], 0.0)
            G56l.append(G5 + G6); aux.append((G, G5, G6))
        grads = [local_grad(s) for s in range(S)]
        fac = None
        mu_aff = 0.0
        for ps in range(2):
            yU = []; yz = []; e12 = []
            for s, sg in enumerate(segs):
                G, G5, G6 = aux[s]; L = loc[s]; t, td = st.t[s], st.td[s]
                rc, rcd, rd, rdd = L["rc"], L["rcd"], L["rd"], L["rdd"]
                c = rc / t
                e1 = L["rsl"] + c[:, 0] + c[:, 2] + G[:, 0] * rd[:, 0] + G[:, 2] * rd[:, 2]
                e2 = L["rsu"] + c[:, 1] + c[:, 3] + G[:, 1] * rd[:, 1] + G[:, 3] * rd[:, 3]
                etal = c[:, 0] + G[:, 0] * rd[:, 0] - G[:, 0] * e1 / (G[:, 0] + G[:, 2])
                etau = -c[:, 1] - G[:, 1] * rd[:, 1] + G[:, 1] * e2 / (G[:, 1] + G[:, 3])
                da = dact(sg)
                ek = np.where(da, (rcd[:, 0] / td[:, 0] + G5 * rdd[:, 0]) - (rcd[:, 1] / td[:, 1] + G6 * rdd[:, 1]), 0.0)
                suf = np.cumsum(ek[::-1])[::-1] - ek
                gU, gz = grads[s]
                yU.append(-(gU + etal + etau + np.where(odd, h * suf[kidx], 0.0)))
                if gz is not None:
                    g2 = gz.copy(); g2[6] += ek.sum()
                    yz.append(-g2)
                else:
                    yz.append(None)
                e12.append((e1, e2))
            dU, dz, fac = newton(Rt, G56l, yU, yz, fac)
            rr = 0.0; steps = []
            for s, sg in enumerate(segs):
                G, G5, G6 = aux[s]; L = loc[s]; t, lam, td, lamd = st.t[s], st.lam[s], st.td[s], st.lamd[s]
                e1, e2 = e12[s]; u = dU[s]; rd, rc, rcd, rdd = L["rd"], L["rc"], L["rcd"], L["rdd"]
                dsl = -(e1 + G[:, 0] * u) / (G[:, 0] + G[:, 2]); dsu = -(e2 - G[:, 1] * u) / (G[:, 1] + G[:, 3])
                dt = np.stack([u + dsl + rd[:, 0], -u + dsu + rd[:, 1], dsl + rd[:, 2], dsu + rd[:, 3]], 1)
                dlam = -rc / t - G * dt
                da = dact(sg)
                u1 = u[1::2]; pre = np.concatenate([[0.0], np.cumsum(u1)[:-1]])
                ddx6 = (0.0 if sg["first"] else dz[s][6]) + h * pre
                dtd = np.where(da[:, None], np.stack([ddx6 + rdd[:, 0], -ddx6 + rdd[:, 1]], 1), 0.0)
                dlamd = np.where(da[:, None], -rcd / td - np.stack([G5, G6], 1) * dtd, 0.0)
                rr = max(rr, (-dt / t).max(), (-dlam / lam).max())
                if da.any():
                    rr = max(rr, (-dtd / td)[da].max(), (-dlamd[da] / lamd[da]).max())
                steps.append((dsl, dsu, dt, dlam, dtd, dlamd))
            amax = 1.0 / rr if rr > 1.0 else 1.0
            if ps == 0:
                s_aff = 0.0
                for s, sg in enumerate(segs):
                    dsl, dsu, dt, dlam, dtd, dlamd = steps[s]; da = dact(sg)
                    s_aff += ((st.t[s] + amax * dt) * (st.lam[s] + amax * dlam)).sum()
                    s_aff += ((st.td[s] + amax * dtd) * (st.lamd[s] + amax * dlamd))[da].sum()
                mu_aff = s_aff / n_ineq
                sigma = (mu_aff / mu) ** 3
                if alpha_prev < BLOCKED_STEP:
                    sigma = 1.0
                smu = max(sigma * mu, MU_FLOOR * cfg.ipm_tol_comp)
                for s, sg in enumerate(segs):
                    dsl, dsu, dt, dlam, dtd, dlamd = steps[s]
                    w2 = 0.0 if cons else 1.0
                    loc[s]["rc"] = st.t[s] * st.lam[s] + w2 * dt * dlam - smu
                    loc[s]["rcd"] = np.where(dact(sg)[:, None], st.td[s] * st.lamd[s] + w2 * dtd * dlamd - smu, 0.0)
            else:
                tau = min(max(1.0 - mu_aff, 0.995), 0.999999)
                alpha = min(tau * amax, 1.0)
                if it == 0 and warmed and alpha < cfg.ipm_warm_restart:
                    warmed = False
                    cold_start()
                    alpha_prev = 1.0; step = 1e300; rmax_prev = 0.0; rstat = -1.0
                    break
                if trace is not None:
                    trace.append((mu, alpha))
                alpha_prev = alpha
                rstat = (1.0 - alpha) * rstat
                step = 0.0
                for s, sg in enumerate(segs):
                    dsl, dsu, dt, dlam, dtd, dlamd = steps[s]; da = dact(sg)
                    step = max(step, np.abs(alpha * dU[s]).max())
                    st.t[s] = np.maximum(st.t[s] + alpha * dt, IPM_FLOOR); st.lam[s] = np.maximum(st.lam[s] + alpha * dlam, IPM_FLOOR)
                    st.U[s] = st.U[s] + alpha * dU[s]; st.sl[s] = st.sl[s] + alpha * dsl; st.su[s] = st.su[s] + alpha * dsu
                    st.z[s] = st.z[s] + alpha * dz[s]
                    st.td[s] = np.where(da[:, None], np.maximum(st.td[s] + alpha * dtd, IPM_FLOOR), 1.0)
                    st.lamd[s] = np.where(da[:, None], np.maximum(st.lamd[s] + alpha * dlamd, IPM_FLOOR), 0.0)
        it += 1
    return _finish(qp, st.U), it


def _finish(qp, Us):
    du = np.concatenate(Us).reshape(qp.N, NU)
    dx = np.empty((qp.N + 1, NX)); dx[0] = qp.dx0
    for k in range(qp.N):
        dx[k + 1] = qp.A[k] @ dx[k] + qp.B[k] @ du[k] + qp.b[k]
    return du, dx
