"""ctypes wrapper of the lane emulator (tests/emu/rowqp_emu.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from ad_mpc_amd.config import AdmpcConfig, NX, NU, NY

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librowqp_emu.so")


def build():
    src = os.path.join(_HERE, "rowqp_emu.cpp")
    core = os.path.join(_HERE, "..", "..", "ad_mpc_amd", "csrc", "rowqp_core.h")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-o", _SO, src], check=True)
    return _SO


def pack_linearisation(oracle, cfg, xbar, ubar, p):
    """GT [B][N][7][6] (stored columns A[:,2..6], B[:,0..1], rows 0..5) and the defects b [B][N][7], as kernel A writes them."""
    B, N = xbar.shape[0], cfg.N
    GT = np.zeros((B, N, 7, 6)); bl = np.zeros((B, N, 7))
    for b in range(B):
        for k in range(N):
            phi, A, Bm = oracle.rk4_sens(cfg, xbar[b, k], ubar[b, k], p[b], cfg.Ts)
            GT[b, k, :5] = A[:6, 2:].T
            GT[b, k, 5:] = Bm[:6, :].T
            bl[b, k] = phi - xbar[b, k + 1]
    return GT, bl


class Emu:
    def __init__(self):
        self.lib = C.CDLL(build())

    def solve(self, cfg, x0, yref, yref_e, GT, bl, xbar, ubar, dtype=np.float64, want_pi=False, split=False):
        N = cfg.N
        c = lambda a: np.ascontiguousarray(a, dtype=dtype)
        x0 = c(x0).reshape(-1, NX); B = x0.shape[0]
        yref = c(yref).reshape(B, N, NY); yref_e = c(yref_e).reshape(B, NX); GT = c(GT).reshape(B, N, 42); bl = c(bl).reshape(B, N, 7)
        x = c(xbar).reshape(B, N + 1, NX).copy(); u = c(ubar).reshape(B, N, NU).copy()
        cost = np.empty(B, dtype=dtype); st = np.empty(B, dtype=np.int32); it = np.empty(B, dtype=np.int32)
        pi = np.zeros((B, N + 1, NX), dtype=dtype) if want_pi else None
        ineq = np.zeros((B, N, 20), dtype=dtype) if want_pi else None
        rmax = np.zeros(B, dtype=dtype)
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)
        if split:               # the device's two-phase path for large batches; self.key: sort keys of the deferred instances
            assert dtype == np.float64
            f = self.lib.rowqp_emu_solve_split_f64
            f.argtypes = [C.POINTER(AdmpcConfig), C.c_int] + [C.c_void_p] * 14
            self.key = np.zeros(B, dtype=np.int32)
            rc = f(C.byref(cfg), B, vp(x0), vp(yref), vp(yref_e), vp(GT), vp(bl), vp(x), vp(u), vp(cost), vp(st), vp(it), vp(pi), vp(ineq), vp(rmax), vp(self.key))
        else:
            f = self.lib.rowqp_emu_solve_f64 if dtype == np.float64 else self.lib.rowqp_emu_solve_f32
            f.argtypes = [C.POINTER(AdmpcConfig), C.c_int] + [C.c_void_p] * 13
            rc = f(C.byref(cfg), B, vp(x0), vp(yref), vp(yref_e), vp(GT), vp(bl), vp(x), vp(u), vp(cost), vp(st), vp(it), vp(pi), vp(ineq), vp(rmax))
        assert rc == 0
        return (x, u, cost, st, it) + ((pi, ineq) if want_pi else ()) + (rmax,)
