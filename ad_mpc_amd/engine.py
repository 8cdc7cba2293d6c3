"""Batch solve engine: thin host wrapper over the C ABI (include/admpc.h).

torch is used only as the carrier of device memory and streams; every number is produced by the
HIP kernels in libadmpc.so.  All tensors are float64 / int32, contiguous, on the solver's device.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import AdmpcConfig, NX, NU, NY, default_config


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class PackedIO:
    """Staging for a single-instance solve through the reference-shaped host classes: every input in ONE pinned host buffer / ONE device buffer
    (each field on a 256-byte boundary, handed to the engine as a contiguous view), every float output behind them, the status words in a pair of
    int32 -- a solve is one copy up, the kernels, one copy down (+ the status pair), one stream synchronisation.  (A tensor per field, each with a
    copy of its own, was 0.2 ms of a 0.58 ms control step of the car.)"""

    def __init__(self, device, fields_in, fields_out, n_int=2):
        self.device = device
        self.shape, self.off, o = {}, {}, 0
        for k, shp in tuple(fields_in) + tuple(fields_out):
            n = int(np.prod(shp)); self.shape[k] = tuple(shp); self.off[k] = (o, n); o += -(-n // 32) * 32
        self.n_in = self.off[fields_out[0][0]][0] if fields_out else o
        self.hbuf = torch.zeros(o, dtype=torch.float64).pin_memory(); self.dbuf = torch.zeros(o, dtype=torch.float64, device=device)
        self.hint = torch.zeros(n_int, dtype=torch.int32).pin_memory(); self.dint = torch.zeros(n_int, dtype=torch.int32, device=device)
        self.h, self.hi = self.hbuf.numpy(), self.hint.numpy()

    def put(self, k, a):
        o, n = self.off[k]; self.h[o:o + n] = np.asarray(a, dtype=np.float64).reshape(-1)

    def dev(self, k):
        o, n = self.off[k]; return self.dbuf[o:o + n].view(self.shape[k])

    def take(self, k):
        o, n = self.off[k]; return self.h[o:o + n].reshape(self.shape[k]).copy()

    def upload(self):
        self.dbuf[:self.n_in].copy_(self.hbuf[:self.n_in], non_blocking=True)

    def download(self, first):
        """Everything from field `first` on comes back (the iterate is updated in place: it sits at the end of the inputs), then one synchronisation."""
        o = self.off[first][0]
        self.hbuf[o:].copy_(self.dbuf[o:], non_blocking=True); self.hint.copy_(self.dint, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()


class BatchSolver:
    """One solver object per (config, device).  Replaces the AcadosOcpSolver object of
    ad_3d_optimizer.py:209 for a whole batch of independent MPC instances."""

    def __init__(self, cfg=None, device=0):
        if not torch.cuda.is_available():
            raise _lib.AdmpcError("no HIP device visible: the AD-MPC engine has no CPU fallback")
        self.lib = _lib.load()
        self.cfg = (cfg if cfg is not None else default_config()).copy()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p(0)
        _lib.check(self.lib.admpc_create(C.byref(self.cfg), self.device_index, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.admpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------------
    @property
    def N(self):
        return int(self.cfg.N)

    def to_device(self, a, dtype=torch.float64):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device).contiguous()

    def _chk(self, t, shape, dtype=torch.float64):
        if t.dtype != dtype or not t.is_contiguous() or t.device != self.device or tuple(t.shape) != tuple(shape):
            raise ValueError("expected contiguous %s tensor of shape %s on %s, got %s %s on %s"
                             % (dtype, tuple(shape), self.device, t.dtype, tuple(t.shape), t.device))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- the hot path --------------------------------------------------------------------------
    def solve(self, x0, yref, yref_e, p, xbar, ubar, cost=None, status=None, iters=None):
        """One SQP-RTI step for every instance, in place on xbar/ubar (device tensors).  Asynchronous.
        The dtype of x0 selects the path: float64 (admpc_solve_batch) or float32 (admpc_solve_batch_f32, storage and compute)."""
        N = self.N
        B = x0.shape[0]
        dt = x0.dtype
        if dt not in (torch.float64, torch.float32):
            raise ValueError("x0 must be float64 or float32")
        self._chk(x0, (B, NX), dt); self._chk(yref, (B, N, NY), dt); self._chk(yref_e, (B, NX), dt); self._chk(p, (B,), dt)
        self._chk(xbar, (B, N + 1, NX), dt); self._chk(ubar, (B, N, NU), dt)
        if cost is not None: self._chk(cost, (B,), dt)
        if status is not None: self._chk(status, (B,), torch.int32)
        if iters is not None: self._chk(iters, (B,), torch.int32)
        fn = self.lib.admpc_solve_batch if dt == torch.float64 else self.lib.admpc_solve_batch_f32
        _lib.check(fn(self._h, B, _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(p), _ptr(xbar), _ptr(ubar),
                      _ptr(cost), _ptr(status), _ptr(iters), self._stream()))

    def solve_with_multipliers(self, x0, yref, yref_e, p, xbar, ubar, cost=None, status=None, iters=None):
        """admpc_solve_batch_ex (float64): the step plus pi [B,N+1,7] and ineq [B,N,20] (see include/admpc.h)."""
        N = self.N
        B = x0.shape[0]
        self._chk(x0, (B, NX)); self._chk(yref, (B, N, NY)); self._chk(yref_e, (B, NX)); self._chk(p, (B,))
        self._chk(xbar, (B, N + 1, NX)); self._chk(ubar, (B, N, NU))
        pi = torch.empty((B, N + 1, NX), dtype=torch.float64, device=self.device)
        ineq = torch.empty((B, N, 20), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.admpc_solve_batch_ex(self._h, B, _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(p), _ptr(xbar), _ptr(ubar),
                                                 _ptr(cost), _ptr(status), _ptr(iters), _ptr(pi), _ptr(ineq), self._stream()))
        return pi, ineq

    def nlp_residuals(self, x0, yref, yref_e, p, xbar, ubar, pi, ineq):
        """admpc_nlp_residuals_batch: [B,4] = (res_stat, res_eq, res_ineq, res_comp) of acados' SQP stopping test at the iterate
        (xbar, ubar) with the multipliers solve_with_multipliers returned for it (AcadosOcpSolver.get_residuals())."""
        N = self.N
        B = x0.shape[0]
        self._chk(x0, (B, NX)); self._chk(yref, (B, N, NY)); self._chk(yref_e, (B, NX)); self._chk(p, (B,))
        self._chk(xbar, (B, N + 1, NX)); self._chk(ubar, (B, N, NU)); self._chk(pi, (B, N + 1, NX)); self._chk(ineq, (B, N, 20))
        res = torch.empty((B, 4), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.admpc_nlp_residuals_batch(self._h, B, _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(p), _ptr(xbar), _ptr(ubar),
                                                      _ptr(pi), _ptr(ineq), _ptr(res), self._stream()))
        return res

    def solve_numpy(self, x0, yref, yref_e, p, xbar, ubar, dtype=np.float64):
        """Convenience for tests / the single-instance shims: host arrays in, host arrays out
        (x, u, cost, status, iters).  dtype=np.float32 runs the fp32 path."""
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        d = lambda a: self.to_device(a, tdt)
        x0 = np.asarray(x0, dtype=dtype).reshape(-1, NX)
        B, N = x0.shape[0], self.N
        tx0 = d(x0); tyr = d(np.asarray(yref, dtype=dtype).reshape(B, N, NY)); tye = d(np.asarray(yref_e, dtype=dtype).reshape(B, NX))
        tp = d(np.asarray(p, dtype=dtype).reshape(B))
        tx = d(np.asarray(xbar, dtype=dtype).reshape(B, N + 1, NX)).clone(); tu = d(np.asarray(ubar, dtype=dtype).reshape(B, N, NU)).clone()
        cost = torch.empty(B, dtype=tdt, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        iters = torch.empty(B, dtype=torch.int32, device=self.device)
        self.solve(tx0, tyr, tye, tp, tx, tu, cost, status, iters)
        torch.cuda.synchronize(self.device)
        return tx.cpu().numpy(), tu.cpu().numpy(), cost.cpu().numpy(), status.cpu().numpy(), iters.cpu().numpy()

    def shoot(self, xbar, ubar, p):
        """H1 only: phi [B,N,7], A [B,N,7,7], B [B,N,7,2] (device tensors)."""
        N = self.N
        B = xbar.shape[0]
        self._chk(xbar, (B, N + 1, NX)); self._chk(ubar, (B, N, NU)); self._chk(p, (B,))
        phi = torch.empty((B, N, NX), dtype=torch.float64, device=self.device)
        A = torch.empty((B, N, NX, NX), dtype=torch.float64, device=self.device)
        Bm = torch.empty((B, N, NX, NU), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.admpc_shoot_batch(self._h, B, _ptr(xbar), _ptr(ubar), _ptr(p), _ptr(phi), _ptr(A), _ptr(Bm), self._stream()))
        return phi, A, Bm

    def shift(self, xbar, ubar, p=None, rollout=True):
        """Receding-horizon shift of the iterate, in place (the reference itself never shifts).
        rollout: new terminal state by one model step under the last input (needs p), else a copy."""
        N = self.N
        B = xbar.shape[0]
        self._chk(xbar, (B, N + 1, NX)); self._chk(ubar, (B, N, NU))
        if rollout:
            if p is None:
                raise ValueError("shift(rollout=True) needs the blend parameter p")
            self._chk(p, (B,))
        _lib.check(self.lib.admpc_shift_batch(self._h, B, _ptr(xbar), _ptr(ubar), _ptr(p if rollout else None), 1 if rollout else 0,
                                              self._stream()))

    def argmin(self, cost, index_offset=0):
        """Local arg-min over a device cost vector -> (val tensor[1], idx tensor[1] int64), ties -> lowest index."""
        B = cost.shape[0]
        self._chk(cost, (B,))
        val = torch.empty(1, dtype=torch.float64, device=self.device)
        idx = torch.empty(1, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.admpc_argmin(self._h, _ptr(cost), B, int(index_offset), _ptr(val), _ptr(idx), self._stream()))
        return val, idx

    def argmin_pair(self, cost, index_offset=0, pair=None):
        """admpc_argmin with value and index written next to each other: a float64[2] tensor whose second element carries
        the bits of the int64 global index -- the 16-byte record that is all-gathered across GPUs."""
        B = cost.shape[0]
        self._chk(cost, (B,))
        if pair is None:
            pair = torch.empty(2, dtype=torch.float64, device=self.device)
        self._chk(pair, (2,))
        _lib.check(self.lib.admpc_argmin(self._h, _ptr(cost), B, int(index_offset), C.c_void_p(pair.data_ptr()),
                                         C.c_void_p(pair.data_ptr() + 8), self._stream()))
        return pair

    def argmin_pairs(self, pairs, out=None):
        """Second level: float64[W,2] gathered records -> float64[2] record of the winner (admpc_argmin_pairs)."""
        W = pairs.shape[0]
        self._chk(pairs, (W, 2))
        if out is None:
            out = torch.empty(2, dtype=torch.float64, device=self.device)
        self._chk(out, (2,))
        _lib.check(self.lib.admpc_argmin_pairs(self._h, _ptr(pairs), W, C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 8),
                                               self._stream()))
        return out

    def epilogue(self, xopt, uopt, xref_xy):
        """Validity bit + Ackermann record per instance."""
        N = self.N
        B = xopt.shape[0]
        self._chk(xopt, (B, N + 1, NX)); self._chk(uopt, (B, N, NU)); self._chk(xref_xy, (B, N + 1, 2))
        ack = torch.empty((B, 4), dtype=torch.float32, device=self.device)
        valid = torch.empty(B, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.admpc_epilogue_batch(self._h, B, _ptr(xopt), _ptr(uopt), _ptr(xref_xy), _ptr(ack), _ptr(valid), self._stream()))
        return ack, valid

    def actuation(self, xopt, uopt, xref_xy, status, steer_meas, safe_count, threshold=10, cost=None):
        """The node's post-solve branch per slot: validity, consecutive-success gate (safe_count is updated in
        place), steering command or brake fallback.  Returns (ack float32 [B,4], mode int32 [B], valid int32 [B]); with `cost`
        given, the cost of every slot that issues no MPC command is set to +inf in place (arg-min over valid candidates)."""
        N = self.N
        B = xopt.shape[0]
        self._chk(xopt, (B, N + 1, NX)); self._chk(uopt, (B, N, NU)); self._chk(xref_xy, (B, N + 1, 2))
        self._chk(status, (B,), torch.int32); self._chk(steer_meas, (B,)); self._chk(safe_count, (B,), torch.int32)
        if cost is not None: self._chk(cost, (B,))
        ack = torch.empty((B, 4), dtype=torch.float32, device=self.device)
        mode = torch.empty(B, dtype=torch.int32, device=self.device)
        valid = torch.empty(B, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.admpc_actuation_batch(self._h, B, _ptr(xopt), _ptr(uopt), _ptr(xref_xy), _ptr(status), _ptr(steer_meas), _ptr(safe_count),
                                                  int(threshold), _ptr(cost), _ptr(ack), _ptr(mode), _ptr(valid), self._stream()))
        return ack, mode, valid

    def resample_vel(self, vel_ref, vx, vy, acc_max, dt):
        """gp_ad_mpc_node.py:344-349 on B rows at once, in place: vel_ref float64 [B,H] (may be a strided row view with unit inner stride)."""
        B, H = vel_ref.shape
        if vel_ref.dtype != torch.float64 or vel_ref.stride(1) != 1 or vel_ref.device != self.device:
            raise ValueError("vel_ref: float64 [B,H] on the solver's device with unit inner stride")
        self._chk(vx, (B,)); self._chk(vy, (B,))
        ld = vel_ref.stride(0) if B > 1 else H
        _lib.check(self.lib.admpc_resample_vel_batch(self.device_index, B, H, int(ld), _ptr(vx), _ptr(vy), float(acc_max), float(dt),
                                                     C.c_void_p(vel_ref.data_ptr()), self._stream()))


class EnsembleBatchSolver:
    """Clustered GP ensembles (reference: one AcadosOcpSolver per cluster, quad_3d_optimizer.py:207, chosen per
    solve by GPEnsemble.select_gp, :452 / :491).  One engine handle per cluster; selection and routing are HIP kernels of the
    library (admpc_select_cluster_batch, admpc_solve_batch_routed): no gather / scatter, no host synchronisation -- every
    handle runs over the whole batch and leaves the instances of the other clusters alone.  ``ensemble``: gp_loader.GPEnsemble."""

    def __init__(self, cfg, ensemble, device=0):
        from .config import set_gp
        self.ensemble = ensemble
        self.solvers = []
        for c in range(ensemble.n_models):
            cc = cfg.copy()
            set_gp(cc, ensemble.clusters[c])
            self.solvers.append(BatchSolver(cc, device=device))
        self.lib = self.solvers[0].lib
        self.device = self.solvers[0].device
        self.device_index = self.solvers[0].device_index
        self.N = self.solvers[0].N
        self._cent = torch.as_tensor(np.ascontiguousarray(ensemble.centroids, dtype=np.float64), device=self.device).contiguous()      # K x d
        feats = [int(f) for f in np.atleast_1d(ensemble.feats)]
        self._feats = (C.c_int32 * len(feats))(*feats)
        self._handles = (C.c_void_p * len(self.solvers))(*[sv._h for sv in self.solvers])

    def close(self):
        for s in self.solvers:
            s.close()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def select(self, x_sel, u_sel, route=None):
        """Cluster of every instance from the state / input the caller selects on (the reference passes the reference state and
        the input target): nearest centroid of the ensemble's feature, ties to the lowest index.  float64 [B,7] / [B,2] device
        tensors in, int32 [B] device tensor out.  Asynchronous."""
        B = x_sel.shape[0]
        sv = self.solvers[0]
        x_sel, u_sel = x_sel.contiguous(), u_sel.contiguous()           # row views (e.g. ubar[:, 0, :]) are welcome
        sv._chk(x_sel, (B, NX)); sv._chk(u_sel, (B, NU))
        if route is None:
            route = torch.empty(B, dtype=torch.int32, device=self.device)
        sv._chk(route, (B,), torch.int32)
        _lib.check(self.lib.admpc_select_cluster_batch(self.device_index, B, len(self._feats), self._feats, _ptr(x_sel), _ptr(u_sel),
                                                       int(self._cent.shape[0]), _ptr(self._cent), _ptr(route), self._stream()))
        return route

    def solve(self, gp_ind, x0, yref, yref_e, p, xbar, ubar, cost=None, status=None, iters=None):
        """BatchSolver.solve with a cluster index per instance (int32 device tensor, e.g. from `select`; int64 is converted).  In place
        on xbar / ubar.  Asynchronous.  An index outside [0, K) gives status 4 and cost +inf for that instance."""
        N, B = self.N, x0.shape[0]
        sv = self.solvers[0]
        if gp_ind.dtype != torch.int32:
            gp_ind = gp_ind.to(torch.int32)
        sv._chk(gp_ind, (B,), torch.int32)
        sv._chk(x0, (B, NX)); sv._chk(yref, (B, N, NY)); sv._chk(yref_e, (B, NX)); sv._chk(p, (B,))
        sv._chk(xbar, (B, N + 1, NX)); sv._chk(ubar, (B, N, NU))
        if cost is not None: sv._chk(cost, (B,))
        if status is not None: sv._chk(status, (B,), torch.int32)
        if iters is not None: sv._chk(iters, (B,), torch.int32)
        _lib.check(self.lib.admpc_solve_batch_routed(self._handles, len(self.solvers), B, _ptr(gp_ind), _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(p),
                                                     _ptr(xbar), _ptr(ubar), _ptr(cost), _ptr(status), _ptr(iters), self._stream()))


class QuadBatchSolver:
    """The second vehicle model (include/admpc_quad.h): one SQP-RTI step of the reference's quadrotor MPC for a batch.
    Replaces the AcadosOcpSolver of quad_3d_optimizer.py:207 for B independent instances."""

    def __init__(self, cfg=None, device=0):
        from .quad_config import default_quad_config
        if not torch.cuda.is_available():
            raise _lib.AdmpcError("no HIP device visible: the AD-MPC engine has no CPU fallback")
        self.lib = _lib.load()
        self.cfg = (cfg if cfg is not None else default_quad_config()).copy()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p(0)
        _lib.check(self.lib.admpc_quad_create(C.byref(self.cfg), self.device_index, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.admpc_quad_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t, shape, dtype=torch.float64):
        if t.dtype != dtype or not t.is_contiguous() or t.device != self.device or tuple(t.shape) != tuple(shape):
            raise ValueError("expected contiguous %s tensor of shape %s on %s" % (dtype, tuple(shape), self.device))

    def solve(self, x0, yref, yref_e, xbar, ubar, cost=None, status=None, iters=None, gp_state=None):
        """In place on xbar [B,N+1,13] / ubar [B,N,4] (float64 device tensors).  Asynchronous.  gp_state [B,13]: the first node's GP
        state (run_optimization's gp_regression_state); None: the initial state x0, the reference's default."""
        from .quad_config import QNX, QNU, QNY
        N, B = int(self.cfg.N), x0.shape[0]
        self._chk(x0, (B, QNX)); self._chk(yref, (B, N, QNY)); self._chk(yref_e, (B, QNX)); self._chk(xbar, (B, N + 1, QNX)); self._chk(ubar, (B, N, QNU))
        if cost is not None: self._chk(cost, (B,))
        if status is not None: self._chk(status, (B,), torch.int32)
        if iters is not None: self._chk(iters, (B,), torch.int32)
        if gp_state is not None: self._chk(gp_state, (B, QNX))
        _lib.check(self.lib.admpc_quad_solve_batch_ex(self._h, B, _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(gp_state), _ptr(xbar), _ptr(ubar),
                                                      _ptr(cost), _ptr(status), _ptr(iters), self._stream()))

    def shoot(self, xbar, ubar, gp_state=None):
        from .quad_config import QNX, QNU
        N, B = int(self.cfg.N), xbar.shape[0]
        self._chk(xbar, (B, N + 1, QNX)); self._chk(ubar, (B, N, QNU))
        if gp_state is not None: self._chk(gp_state, (B, QNX))
        phi = torch.empty((B, N, QNX), dtype=torch.float64, device=self.device)
        A = torch.empty((B, N, QNX, QNX), dtype=torch.float64, device=self.device)
        Bm = torch.empty((B, N, QNX, QNU), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.admpc_quad_shoot_batch_ex(self._h, B, _ptr(xbar), _ptr(ubar), _ptr(gp_state), _ptr(phi), _ptr(A), _ptr(Bm), self._stream()))
        return phi, A, Bm

    def solve_numpy(self, x0, yref, yref_e, xbar, ubar, gp_state=None):
        """Host arrays in, host arrays out: (x, u, cost, status, iters)."""
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=self.device)
        B = np.asarray(x0).shape[0]
        tx, tu = d(xbar).clone(), d(ubar).clone()
        cost = torch.empty(B, dtype=torch.float64, device=self.device); st = torch.empty(B, dtype=torch.int32, device=self.device); it = torch.empty(B, dtype=torch.int32, device=self.device)
        self.solve(d(x0), d(yref), d(yref_e), tx, tu, cost, st, it, gp_state=None if gp_state is None else d(gp_state))
        torch.cuda.synchronize(self.device)
        return tx.cpu().numpy(), tu.cpu().numpy(), cost.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()


class QuadEnsembleBatchSolver:
    """Clustered GP ensembles of the quadrotor (reference: one AcadosOcpSolver per cluster, quad_3d_optimizer.py:207, chosen per solve
    from the reference state by GPEnsemble.select_gp, :446-452 / :485-491).  One handle per cluster; selection and routing are HIP
    kernels of the library (admpc_quad_select_cluster_batch, admpc_quad_solve_batch_routed).  ``clusters[c]``: the ``set_quad_gp`` list
    of cluster c; ``centroids`` K x d in the features ``feats`` (indices into z = [x with the velocity in the body frame; u])."""

    def __init__(self, cfg, clusters, centroids, feats, device=0):
        from .quad_config import set_quad_gp
        self.solvers = []
        for gps in clusters:
            cc = cfg.copy()
            set_quad_gp(cc, gps)
            self.solvers.append(QuadBatchSolver(cc, device=device))
        s0 = self.solvers[0]
        self.lib, self.device, self.device_index, self.N = s0.lib, s0.device, s0.device_index, int(s0.cfg.N)
        self._cent = torch.as_tensor(np.ascontiguousarray(np.asarray(centroids, dtype=np.float64).reshape(len(clusters), -1)), device=self.device).contiguous()
        feats = [int(f) for f in np.atleast_1d(feats)]
        if self._cent.shape[1] != len(feats):
            raise ValueError("centroids must be K x len(feats)")
        self._feats = (C.c_int32 * len(feats))(*feats)
        self._handles = (C.c_void_p * len(self.solvers))(*[sv._h for sv in self.solvers])

    def close(self):
        for s in self.solvers:
            s.close()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def select(self, x_sel, u_sel, route=None):
        """Cluster of every instance from the reference state [B,13] (world-frame velocity: the kernel rotates it to the body frame as
        the reference does before select_gp) and input target [B,4]; int32 [B] device tensor out.  Asynchronous."""
        from .quad_config import QNX, QNU
        B = x_sel.shape[0]
        sv = self.solvers[0]
        x_sel, u_sel = x_sel.contiguous(), u_sel.contiguous()
        sv._chk(x_sel, (B, QNX)); sv._chk(u_sel, (B, QNU))
        if route is None:
            route = torch.empty(B, dtype=torch.int32, device=self.device)
        sv._chk(route, (B,), torch.int32)
        _lib.check(self.lib.admpc_quad_select_cluster_batch(self.device_index, B, len(self._feats), self._feats, _ptr(x_sel), _ptr(u_sel),
                                                            int(self._cent.shape[0]), _ptr(self._cent), _ptr(route), self._stream()))
        return route

    def solve(self, gp_ind, x0, yref, yref_e, xbar, ubar, cost=None, status=None, iters=None, gp_state=None):
        """QuadBatchSolver.solve with a cluster index per instance (int32 device tensor).  In place on xbar / ubar.  Asynchronous.
        An index outside [0, K) gives status 4 and cost +inf for that instance."""
        from .quad_config import QNX, QNU, QNY
        N, B = self.N, x0.shape[0]
        sv = self.solvers[0]
        if gp_ind.dtype != torch.int32:
            gp_ind = gp_ind.to(torch.int32)
        sv._chk(gp_ind, (B,), torch.int32)
        sv._chk(x0, (B, QNX)); sv._chk(yref, (B, N, QNY)); sv._chk(yref_e, (B, QNX)); sv._chk(xbar, (B, N + 1, QNX)); sv._chk(ubar, (B, N, QNU))
        if cost is not None: sv._chk(cost, (B,))
        if status is not None: sv._chk(status, (B,), torch.int32)
        if iters is not None: sv._chk(iters, (B,), torch.int32)
        if gp_state is not None: sv._chk(gp_state, (B, QNX))
        _lib.check(self.lib.admpc_quad_solve_batch_routed(self._handles, len(self.solvers), B, _ptr(gp_ind), _ptr(x0), _ptr(yref), _ptr(yref_e), _ptr(gp_state),
                                                          _ptr(xbar), _ptr(ubar), _ptr(cost), _ptr(status), _ptr(iters), self._stream()))
