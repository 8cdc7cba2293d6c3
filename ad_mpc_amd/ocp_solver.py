"""``AcadosOcpSolver``-shaped adapter over the batch engine (narrow seam of SURVEY 8b).

The reference stores an ``acados_template.AcadosOcpSolver`` at ad_3d_optimizer.py:209 and drives it
with ``set(stage, field, value)`` (:330-331,:430,:438,:441-442,:450), ``solve()`` (:456) and
``get(stage, field)`` (:462-465).  This class offers exactly those calls for ONE instance and runs
the solve on the GPU through admpc_solve_batch (B = 1).  Like the acados object it owns a persistent
iterate (initially all zeros, acados_solver_sim_car.c:705-731) that is never shifted or reset --
unless the caller sets ``shift_iterate`` (off by default: an option the reference does not have).
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib
from .config import NX, NU, NY
from .engine import BatchSolver


class AdmpcOcpSolver:
    def __init__(self, cfg, device=0):
        self._eng = BatchSolver(cfg, device=device)
        self.N = N = self._eng.N
        self._yref = np.zeros((N, NY))
        self._yref_e = np.zeros(NX)
        self._lbx0 = np.zeros(NX)
        self._ubx0 = np.zeros(NX)
        self._p = np.zeros(N + 1)
        self._x = np.zeros((N + 1, NX))          # persistent iterate
        self._u = np.zeros((N, NU))
        self._status = 0
        self._qp_iter = 0
        self._cost = float("nan")
        self.shift_iterate = None                # None: keep the iterate as the reference does; "copy" / "rollout": shift before each solve
        # One staging buffer on either side: a solve is ONE host-to-device copy (the measured state, the references, p, the iterate), the
        # kernels, ONE copy back (iterate, cost, multipliers) plus the two status words -- pinned host memory, asynchronous on the solver's
        # stream, one stream synchronisation.  (Six tensors up and seven down, each its own copy, were 0.2 ms of a 0.58 ms control step.)
        fields = (("x0", NX), ("yref", N * NY), ("yref_e", NX), ("p", 1), ("x", (N + 1) * NX), ("u", N * NU), ("cost", 1), ("pi", (N + 1) * NX), ("ineq", N * 20))
        self._off, o = {}, 0
        for k, n in fields:
            self._off[k] = (o, n); o += -(-n // 32) * 32          # every field on a 256-byte boundary, like an allocation of its own
        dev = self._eng.device
        self._hbuf = torch.zeros(o, dtype=torch.float64).pin_memory(); self._dbuf = torch.zeros(o, dtype=torch.float64, device=dev)
        self._hint = torch.zeros(2, dtype=torch.int32).pin_memory(); self._dint = torch.zeros(2, dtype=torch.int32, device=dev)
        self._hnp, self._hint_np = self._hbuf.numpy(), self._hint.numpy()

    # ---- acados-style setters / getters -------------------------------------------------------
    def set(self, stage_, field_, value_):
        value = np.asarray(value_, dtype=np.float64).reshape(-1)
        N = self.N
        if not isinstance(stage_, (int, np.integer)) or not (0 <= stage_ <= N):
            raise Exception("AdmpcOcpSolver.set(): stage index must be an integer in [0, %d], got %r" % (N, stage_))

        def need(n):
            if value.shape[0] != n:
                raise Exception("AdmpcOcpSolver.set(): mismatching dimension for field \"%s\" with dimension %d (you have %d)"
                                % (field_, n, value.shape[0]))
        if field_ == "yref":
            if stage_ < N:
                need(NY); self._yref[stage_] = value
            else:
                need(NX); self._yref_e[:] = value
        elif field_ in ("lbx", "ubx"):
            if stage_ != 0:
                raise Exception("AdmpcOcpSolver.set(): \"%s\" is only settable at stage 0 (initial state)" % field_)
            need(NX)
            (self._lbx0 if field_ == "lbx" else self._ubx0)[:] = value
        elif field_ == "p":
            need(1); self._p[stage_] = value[0]
        elif field_ == "x":
            need(NX); self._x[stage_] = value
        elif field_ == "u":
            if stage_ >= N:
                raise Exception("AdmpcOcpSolver.set(): no input at the terminal stage")
            need(NU); self._u[stage_] = value
        else:
            raise Exception("AdmpcOcpSolver.set(): \"%s\" is not a valid argument. Possible values are "
                            "['yref', 'lbx', 'ubx', 'p', 'x', 'u']" % field_)

    def get(self, stage_, field_):
        N = self.N
        if not isinstance(stage_, (int, np.integer)) or not (0 <= stage_ <= N):
            raise Exception("AdmpcOcpSolver.get(): stage index must be an integer in [0, %d], got %r" % (N, stage_))
        if field_ == "x":
            return self._x[stage_].copy()
        if field_ == "u":
            if stage_ >= N:
                raise Exception("AdmpcOcpSolver.get(): no input at the terminal stage")
            return self._u[stage_].copy()
        raise Exception("AdmpcOcpSolver.get(): \"%s\" is not a valid argument. Possible values are ['x', 'u']" % field_)

    def get_stats(self, field_):
        if field_ == "qp_iter":
            return self._qp_iter
        if field_ == "status":
            return self._status
        raise Exception("AdmpcOcpSolver.get_stats(): unknown field %r" % field_)

    def get_cost(self):
        return self._cost

    # ---- solve --------------------------------------------------------------------------------
    def solve(self):
        """One SQP_RTI step (or cfg.sqp_iters full steps); returns the acados-style status int."""
        if not np.array_equal(self._lbx0, self._ubx0):
            raise Exception("AdmpcOcpSolver.solve(): stage-0 lbx and ubx must both equal the measured state")
        if not np.all(self._p == self._p[0]):
            raise Exception("AdmpcOcpSolver.solve(): the blend parameter p must be the same on all stages")
        if self.shift_iterate not in (None, "copy", "rollout"):
            raise Exception("AdmpcOcpSolver.solve(): shift_iterate must be None, \"copy\" or \"rollout\"")
        if self.shift_iterate is not None:
            d = self._eng.to_device
            tx, tu, tp = d(self._x[None]).clone(), d(self._u[None]).clone(), d(np.array([self._p[0]]))
            self._eng.shift(tx, tu, tp, rollout=self.shift_iterate == "rollout")
            self._x, self._u = tx[0].cpu().numpy(), tu[0].cpu().numpy()
        h, off = self._hnp, self._off
        def put(k, a):
            o, n = off[k]; h[o:o + n] = np.asarray(a, dtype=np.float64).reshape(-1)
        put("x0", self._lbx0); put("yref", self._yref); put("yref_e", self._yref_e); h[off["p"][0]] = self._p[0]; put("x", self._x); put("u", self._u)
        n_in, o_out = off["cost"][0], off["x"][0]
        eng = self._eng
        with torch.cuda.device(eng.device):
            stream = torch.cuda.current_stream(eng.device)
            self._dbuf[:n_in].copy_(self._hbuf[:n_in], non_blocking=True)
            base = self._dbuf.data_ptr()
            P = lambda k: C.c_void_p(base + 8 * off[k][0])
            ip = self._dint.data_ptr()
            _lib.check(eng.lib.admpc_solve_batch_ex(eng._h, 1, P("x0"), P("yref"), P("yref_e"), P("p"), P("x"), P("u"), P("cost"),
                                                    C.c_void_p(ip), C.c_void_p(ip + 4), P("pi"), P("ineq"), C.c_void_p(stream.cuda_stream)))
            self._hbuf[o_out:].copy_(self._dbuf[o_out:], non_blocking=True)
            self._hint.copy_(self._dint, non_blocking=True)
            stream.synchronize()
        def take(k, shape):
            o, n = off[k]; return h[o:o + n].reshape(shape).copy()
        st, it, cost = int(self._hint_np[0]), int(self._hint_np[1]), float(h[off["cost"][0]])
        if st in (0, 2):     # acados leaves the iterate untouched only if the QP failed outright; status 2 (SQP limit) keeps the last iterate
            N = self.N
            self._x, self._u = take("x", (N + 1, NX)), take("u", (N, NU))
            self._pi, self._ineq = take("pi", (N + 1, NX)), take("ineq", (N, 20))
        self._status, self._qp_iter, self._cost = st, it, cost
        return self._status

    # ---- iterate snapshots (acados store_iterate / load_iterate JSON format, e.g. sim_car_iterate.json)
    def store_iterate(self, filename="", overwrite=False):
        if filename == "":
            filename = "admpc_iterate.json"
        if not overwrite and os.path.isfile(filename):
            raise Exception("AdmpcOcpSolver.store_iterate(): file %s exists (use overwrite=True)" % filename)
        N = self.N
        pi = getattr(self, "_pi", None)
        iq = getattr(self, "_ineq", None)
        d = {}
        for k in range(N + 1):
            d["x_%d" % k] = self._x[k].tolist()
            d["u_%d" % k] = self._u[k].tolist() if k < N else []
            d["z_%d" % k] = []
            if pi is None or k == N:                     # no solve yet (multipliers all zero in acados too) / terminal stage: no constraints
                zero = pi is None and k < N
                nb = 7 if k == 0 else 1
                d["pi_%d" % k] = [0.0] * NX if zero else []
                d["lam_%d" % k] = [0.0] * (8 + 2 * nb) if zero else []
                d["t_%d" % k] = [0.0] * (8 + 2 * nb) if zero else []
                d["sl_%d" % k] = [0.0] * NU if zero else []
                d["su_%d" % k] = [0.0] * NU if zero else []
                continue
            t, lam = iq[k, :10], iq[k, 10:]
            # acados order per stage: [lbu(2), lbx(nbx), ubu(2), ubx(nbx), ls(2), us(2)], nbx = 7 at stage 0 (initial-state equality, whose
            # multiplier is split by sign) and 1 (steering angle) afterwards (SURVEY 8c pin 2 (iii)); record order see include/admpc.h
            if k == 0:
                nu0 = pi[N]
                lbx_l, ubx_l = np.maximum(nu0, 0.0).tolist(), np.maximum(-nu0, 0.0).tolist()
                lbx_t, ubx_t = [0.0] * NX, [0.0] * NX
            else:
                lbx_l, ubx_l, lbx_t, ubx_t = [lam[4]], [lam[5]], [t[4]], [t[5]]
            d["pi_%d" % k] = pi[k].tolist()
            d["lam_%d" % k] = [lam[0], lam[2]] + lbx_l + [lam[1], lam[3]] + ubx_l + [lam[6], lam[8]] + [lam[7], lam[9]]
            d["t_%d" % k] = [t[0], t[2]] + lbx_t + [t[1], t[3]] + ubx_t + [t[6], t[8]] + [t[7], t[9]]
            d["sl_%d" % k] = [t[6], t[8]]
            d["su_%d" % k] = [t[7], t[9]]
        d = {k: [float(v) for v in vals] for k, vals in d.items()}
        with open(filename, "w") as f:
            json.dump(d, f, indent=4, sort_keys=True)

    def load_iterate(self, filename):
        if not os.path.isfile(filename):
            raise Exception("AdmpcOcpSolver.load_iterate(): file %s does not exist" % filename)
        with open(filename) as f:
            d = json.load(f)
        for k in range(self.N + 1):
            self.set(k, "x", d["x_%d" % k])
            if k < self.N:
                self.set(k, "u", d["u_%d" % k])
