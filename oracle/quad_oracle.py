"""ctypes wrappers of oracle/libquad_oracle.so (CPU restatement of the quadrotor RTI step) and of oracle/_ref/libquad_ref.so (the
reference's own CasADi-generated quadrotor model, compiled in place by `make -C oracle ref`).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from ad_mpc_amd.quad_config import AdmpcQuadConfig, QNX, QNU, QNY

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _ptr(a, ty=_dp):
    return a.ctypes.data_as(ty) if a is not None else None


class QuadOracle:
    def __init__(self):
        subprocess.run(["make", "-C", _HERE, "libquad_oracle.so"], check=True, stdout=subprocess.DEVNULL)
        self.lib = L = C.CDLL(os.path.join(_HERE, "libquad_oracle.so"))
        cp = C.POINTER(AdmpcQuadConfig)
        L.quad_oracle_f.argtypes = [cp, _dp, _dp, _dp, _dp]
        L.quad_oracle_rk4_sens.argtypes = [cp, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp]
        L.quad_oracle_solve_batch.argtypes = [cp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int]
        L.quad_oracle_qp_debug.argtypes = [cp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]
        L.quad_oracle_nlp_residuals.argtypes = [cp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]

    def f(self, cfg, x, u, gpx=None):
        """gpx: the GP-state parameter of a first node (features and rotation of the GP residual come from it), or None."""
        x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64); out = np.empty(QNX)
        g = None if gpx is None else np.ascontiguousarray(gpx, dtype=np.float64)
        self.lib.quad_oracle_f(C.byref(cfg), _ptr(x), _ptr(u), _ptr(g) if g is not None else None, _ptr(out))
        return out

    def rk4_sens(self, cfg, x, u, h, gpx=None):
        x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        g = None if gpx is None else np.ascontiguousarray(gpx, dtype=np.float64)
        phi = np.empty(QNX); A = np.empty((QNX, QNX)); B = np.empty((QNX, QNU))
        self.lib.quad_oracle_rk4_sens(C.byref(cfg), _ptr(x), _ptr(u), _ptr(g) if g is not None else None, float(h), _ptr(phi), _ptr(A), _ptr(B))
        return phi, A, B

    def solve_batch(self, cfg, x0, yref, yref_e, xbar, ubar, nthreads=1, gp_state=None):
        """Returns (x, u, cost, status, iters); the arguments are not modified.  gp_state [B,13]: the first node's GP state
        (run_optimization's gp_regression_state); None: the initial state."""
        N = cfg.N
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, QNX); B = x0.shape[0]
        yref = np.ascontiguousarray(yref, dtype=np.float64).reshape(B, N, QNY); yref_e = np.ascontiguousarray(yref_e, dtype=np.float64).reshape(B, QNX)
        x = np.array(xbar, dtype=np.float64).reshape(B, N + 1, QNX).copy(); u = np.array(ubar, dtype=np.float64).reshape(B, N, QNU).copy()
        cost = np.empty(B); st = np.empty(B, dtype=np.int32); it = np.empty(B, dtype=np.int32)
        gs = None if gp_state is None else np.ascontiguousarray(gp_state, dtype=np.float64).reshape(B, QNX)
        self.lib.quad_oracle_solve_batch(C.byref(cfg), B, _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(gs) if gs is not None else None, _ptr(x), _ptr(u), _ptr(cost), _ptr(st, _ip), _ptr(it, _ip), int(nthreads))
        return x, u, cost, st, it

    def nlp_residuals(self, cfg, x0, yref, yref_e, xbar, ubar, pi, m):
        """(res_stat, res_eq, res_ineq, res_comp) of one instance's iterate with the multipliers pi [N,13], m [N,4] (lower minus upper)."""
        a = lambda v: np.ascontiguousarray(v, dtype=np.float64)
        x0, yref, yref_e, xbar, ubar, pi, m = map(a, (x0, yref, yref_e, xbar, ubar, pi, m))
        res = np.empty(4)
        self.lib.quad_oracle_nlp_residuals(C.byref(cfg), _ptr(x0), _ptr(yref), _ptr(yref_e), None, _ptr(xbar), _ptr(ubar), _ptr(pi), _ptr(m), _ptr(res))
        return res

    def qp_debug(self, cfg, x0, yref, yref_e, xbar, ubar):
        N = cfg.N; n = N * QNU
        a = lambda v, s: np.array(v, dtype=np.float64).reshape(s).copy()
        x0 = a(x0, (QNX,)); yref = a(yref, (N, QNY)); yref_e = a(yref_e, (QNX,)); x = a(xbar, (N + 1, QNX)); u = a(ubar, (N, QNU))
        H = np.empty((n, n)); g = np.empty(n); it = C.c_int32(0)
        st = self.lib.quad_oracle_qp_debug(C.byref(cfg), _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(x), _ptr(u), _ptr(H), _ptr(g), C.byref(it))
        return dict(H=H, g=g, x=x, u=u, status=st, iters=it.value)


class RefQuadModel:
    """my_quad_expl_ode_fun / my_quad_expl_vde_forw of the reference (src/quad_mpc/c_generated_code/my_quad_model), CasADi calling
    convention: inputs x[13], Sx[13x13] (column-major), Su[13x4], u[4], p (empty); outputs xdot[13], dSx (rows 0..11 of every
    column: the yaw-rate row is structurally zero because J_x = J_y), dSu[13x4]."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libquad_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle ref` in the development container)")
        self.lib = C.CDLL(path)
        sig = [C.POINTER(_dp), C.POINTER(_dp), C.POINTER(C.c_longlong), _dp, C.c_int]
        self.lib.my_quad_expl_ode_fun.argtypes = sig
        self.lib.my_quad_expl_vde_forw.argtypes = sig
        self._iw = (C.c_longlong * 64)()
        self._w = (C.c_double * 8192)()
        self._p = np.zeros(1)

    def ode(self, x, u):
        x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64); out = np.empty(QNX)
        arg = (_dp * 16)(_ptr(x), _ptr(u), _ptr(self._p)); res = (_dp * 16)(_ptr(out))
        self.lib.my_quad_expl_ode_fun(arg, res, self._iw, self._w, 0)
        return out

    def vde_forw(self, x, Sx, Su, u):
        x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        sxf = np.ascontiguousarray(np.asarray(Sx, dtype=np.float64).T.reshape(-1)); suf = np.ascontiguousarray(np.asarray(Su, dtype=np.float64).T.reshape(-1))
        xdot = np.empty(QNX); dsx = np.zeros(QNX * 12); dsu = np.zeros(QNX * QNU)
        arg = (_dp * 16)(_ptr(x), _ptr(sxf), _ptr(suf), _ptr(u), _ptr(self._p)); res = (_dp * 16)(_ptr(xdot), _ptr(dsx), _ptr(dsu))
        self.lib.my_quad_expl_vde_forw(arg, res, self._iw, self._w, 0)
        dSx = np.zeros((QNX, QNX)); dSx[:12, :] = dsx.reshape(QNX, 12).T
        dSu = dsu.reshape(QNU, QNX).T.copy()
        return xdot, dSx, dSu

    def rk4_sens(self, x, u, h):
        """acados ERK4 (one step) on the augmented state using the reference's VDE."""
        x = np.asarray(x, dtype=np.float64)
        cst = [0.0, 0.5, 0.5, 1.0]; wst = [1 / 6, 2 / 6, 2 / 6, 1 / 6]
        kx = np.zeros(QNX); kS = np.zeros((QNX, QNX)); kU = np.zeros((QNX, QNU))
        ax = np.zeros(QNX); aS = np.zeros((QNX, QNX)); aU = np.zeros((QNX, QNU))
        for s in range(4):
            xd, dS, dU = self.vde_forw(x + cst[s] * h * kx, np.eye(QNX) + cst[s] * h * kS, cst[s] * h * kU, u)
            kx, kS, kU = xd, dS, dU
            ax += wst[s] * xd; aS += wst[s] * dS; aU += wst[s] * dU
        return x + h * ax, np.eye(QNX) + h * aS, h * aU
