"""Local reference generator with the interface of the reference's ``RefTrajectory``
(data_driven_mpc/ros_gp_mpc/src/ad_mpc/ref_traj.py:41-171) -- SURVEY 8f-1, the step immediately before the solve.

``set_traj`` (once per global path; host, numpy/scipy exactly as the reference) builds the trajectory table;
``get_waypoints`` (every pose message) runs on the GPU through ``admpc_waypoints_batch`` -- for one pose like the
reference, or for a whole batch of poses with ``get_waypoints_batch`` so that scenario references are produced on the
device instead of being streamed from the host.
"""
import ctypes as C
import math

import numpy as np
import torch
from scipy.signal import filtfilt

from . import _lib


def compute_curvature(cdists, psis):
    """Finite-difference curvature + the 11-tap zero-phase moving average of ref_traj.py:10-25."""
    diff_dists = np.diff(cdists)
    diff_psis = np.diff(np.unwrap(psis))
    assert np.max(np.abs(diff_psis)) < np.pi, "Detected a jump in the angle difference."
    curv_raw = diff_psis / np.maximum(diff_dists, 0.1)
    curv_raw = np.insert(curv_raw, len(curv_raw), curv_raw[-1])
    return filtfilt(np.ones((11,)) / 11, 1, curv_raw)


def bound_angle_within_pi(angle):
    return (angle + np.pi) % (2.0 * np.pi) - np.pi          # ref_traj.py:29-30


class RefTrajectory:
    KEYS = ("vel", "x", "y", "psi", "cdist", "curv")

    def __init__(self, traj_horizon=10, traj_dt=0.2, device=0):
        self.traj_horizon = traj_horizon
        self.traj_dt = traj_dt
        self.x_ref, self.y_ref, self.psi_ref, self.vel_ref = [], [], [], []
        self.trajectory = []
        self.access_map = []
        self._dev = torch.device("cuda", int(device))
        self._dev_index = int(device)
        self._cols = None

    def set_traj(self, x_ref, y_ref, psi_ref, vel_ref):
        """ref_traj.py:67-86: cumulative arc length, curvature, table [vel, x, y, psi, cdist, curv]."""
        x_ref = np.asarray(x_ref, dtype=np.float64); y_ref = np.asarray(y_ref, dtype=np.float64)
        seg = np.sqrt(np.diff(x_ref) ** 2 + np.diff(y_ref) ** 2)
        cdists = [0.0]
        for d in seg:                                         # running sum in the reference's order
            cdists.append(d + cdists[-1])
        curvs = compute_curvature(cdists, psi_ref)
        self.trajectory = np.column_stack((vel_ref, x_ref, y_ref, psi_ref, cdists, curvs))
        self.access_map = {key: index for index, key in enumerate(self.KEYS)}
        if not torch.cuda.is_available():
            raise _lib.AdmpcError("no HIP device visible: the reference generator has no CPU fallback")
        t = self.trajectory
        cols = [t[:, 0], t[:, 1], t[:, 2], t[:, 3], np.unwrap(t[:, 3]), t[:, 4], t[:, 5]]
        self._cols = [torch.as_tensor(np.ascontiguousarray(c), dtype=torch.float64, device=self._dev) for c in cols]

    # ---- batched device path --------------------------------------------------------------------
    def get_waypoints_batch(self, X_init, Y_init, psi_init):
        """Device tensors in (shape [B]) -> (ref [B,6,H] = x,y,psi,v,cdist,curv ; err [B,3] = s0,e_y0,e_psi0 ; stop [B] int32)."""
        if self._cols is None:
            raise _lib.AdmpcError("trajectory has not been set")
        L = _lib.load()
        B = int(X_init.shape[0]); H = int(self.traj_horizon); M = int(self.trajectory.shape[0])
        ref = torch.empty((B, 6, H), dtype=torch.float64, device=self._dev)
        err = torch.empty((B, 3), dtype=torch.float64, device=self._dev)
        stop = torch.empty(B, dtype=torch.int32, device=self._dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(L.admpc_waypoints_batch(self._dev_index, M, H, float(self.traj_dt), B, *[p(c) for c in self._cols],
                                           p(X_init), p(Y_init), p(psi_init), p(ref), p(err), p(stop), stream))
        return ref, err, stop

    # ---- the reference's single-pose call ---------------------------------------------------------
    def get_waypoints(self, X_init, Y_init, psi_init):
        """Same dictionary as ref_traj.py:89-171 for one pose."""
        d = lambda v: torch.tensor([float(v)], dtype=torch.float64, device=self._dev)
        ref, err, stop = self.get_waypoints_batch(d(X_init), d(Y_init), d(psi_init))
        torch.cuda.synchronize(self._dev)
        ref = ref.cpu().numpy()[0]; err = err.cpu().numpy()[0]
        return {"s0": err[0], "e_y0": err[1], "e_psi0": err[2], "x_ref": ref[0], "y_ref": ref[1], "psi_ref": ref[2],
                "v_ref": ref[3], "cdist_ref": ref[4], "curv_ref": ref[5], "stop": bool(stop.cpu().numpy()[0])}
