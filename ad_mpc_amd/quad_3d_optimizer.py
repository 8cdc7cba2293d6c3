"""Host mirror of the reference's ``Quad3DOptimizer`` (src/quad_mpc/quad_3d_optimizer.py) over the quadrotor path of the engine
(include/admpc_quad.h): same constructor arguments that shape the OCP, ``set_reference_state`` / ``set_reference_trajectory`` /
``run_optimization`` with the reference's quirks (weight expansion for the quaternion, body-frame velocity in the point reference,
last-row padding of short trajectories, persistent never-shifted iterate).  One vehicle = a batch of one."""
from copy import copy

import numpy as np

from .quad_config import default_quad_config, QNX, QNU


def q_to_rot_mat(q):
    """src/utils/utils.py:323-338."""
    qw, qx, qy, qz = q
    return np.array([[1 - 2 * (qy ** 2 + qz ** 2), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                     [2 * (qx * qy + qw * qz), 1 - 2 * (qx ** 2 + qz ** 2), 2 * (qy * qz - qw * qx)],
                     [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx ** 2 + qy ** 2)]])


def v_dot_q(v, q):
    return q_to_rot_mat(np.asarray(q, dtype=np.float64)).dot(np.asarray(v, dtype=np.float64))


def quaternion_inverse(q):
    return np.array([q[0], -q[1], -q[2], -q[3]], dtype=np.float64)


class Quad3DOptimizer:
    def __init__(self, quad=None, t_horizon=1, n_nodes=20, q_cost=None, r_cost=None, q_mask=None, solver_options=None, device=0):
        """quad_3d_optimizer.py:40-207.  ``quad``: an object with mass, J, max_thrust, x_f, y_f, z_l_tau, min_u / max_u (or
        max_input_value / min_input_value) as ``Quadrotor3D`` has them; None = the vehicle of quad_3d.py."""
        from .engine import QuadBatchSolver
        if q_cost is None:
            q_cost = np.array([10, 10, 10, 0.1, 0.1, 0.1, 0.05, 0.05, 0.05, 0.05, 0.05, 0.05])
        if r_cost is None:
            r_cost = np.array([0.1, 0.1, 0.1, 0.1])
        self.T, self.N = float(t_horizon), int(n_nodes)
        cfg = default_quad_config(N=self.N, t_horizon=self.T)
        q_cost = np.asarray(q_cost, dtype=np.float64)
        # :140-143: one more weight for the rotation (mean of the three angle weights), masked out with the rest of the mask
        q_diagonal = np.concatenate((q_cost[:3], np.mean(q_cost[3:6])[np.newaxis], q_cost[3:]))
        if q_mask is not None:
            q_mask = np.asarray(q_mask, dtype=np.float64)
            q_diagonal = q_diagonal * np.concatenate((q_mask[:3], np.zeros(1), q_mask[3:]))
        terminal_cost = 0 if solver_options is None or not solver_options["terminal_cost"] else 1
        for i in range(QNX):
            cfg.W[i] = float(q_diagonal[i]); cfg.We[i] = float(q_diagonal[i]) * terminal_cost
        for m in range(QNU):
            cfg.W[QNX + m] = float(np.asarray(r_cost)[m])
        if quad is not None:
            cfg.mass, cfg.max_thrust = float(quad.mass), float(quad.max_thrust)
            for i in range(3):
                cfg.J[i] = float(quad.J[i])
            for i in range(4):
                cfg.x_f[i], cfg.y_f[i], cfg.z_l_tau[i] = float(quad.x_f[i]), float(quad.y_f[i]), float(quad.z_l_tau[i])
            lo = getattr(quad, "min_u", getattr(quad, "min_input_value", 0.0)); hi = getattr(quad, "max_u", getattr(quad, "max_input_value", 1.0))
            for m in range(QNU):
                cfg.lbu[m], cfg.ubu[m] = float(lo), float(hi)
        self.cfg = cfg
        self.solver = QuadBatchSolver(cfg, device=device)
        self.yref = np.zeros((self.N, QNX + QNU)); self.yref_e = np.zeros(QNX)
        self.x_iter = np.zeros((self.N + 1, QNX)); self.u_iter = np.zeros((self.N, QNU))      # acados starts from a zero iterate and keeps it
        self.target = None
        self.status = 0

    def set_reference_state(self, x_target=None, u_target=None):
        """:430-463.  The velocity of the point reference goes to the body frame (the reference does that to the yref it sets)."""
        if x_target is None:
            x_target = [[0, 0, 0], [1, 0, 0, 0], [0, 0, 0], [0, 0, 0]]
        if u_target is None:
            u_target = [0, 0, 0, 0]
        self.target = copy(x_target)
        ref = np.concatenate([np.asarray(x_target[i], dtype=np.float64) for i in range(4)])
        v_b = v_dot_q(ref[7:10], quaternion_inverse(ref[3:7]))
        ref = np.concatenate((ref[:7], v_b, ref[10:], np.asarray(u_target, dtype=np.float64)))
        self.yref[:] = ref
        self.yref_e[:] = ref[:-4]
        return 0

    def set_reference_trajectory(self, x_target, u_target):
        """:465-503: short trajectories are padded with their last row; the last node has a state reference only."""
        if u_target is not None:
            assert x_target[0].shape[0] == (u_target.shape[0] + 1) or x_target[0].shape[0] == u_target.shape[0]
        while x_target[0].shape[0] < self.N + 1:
            x_target = [np.concatenate((x, np.expand_dims(x[-1, :], 0)), 0) for x in x_target]
            if u_target is not None:
                u_target = np.concatenate((u_target, np.expand_dims(u_target[-1, :], 0)), 0)
        stacked = np.concatenate([x for x in x_target], 1)
        self.target = copy(x_target)
        for j in range(self.N):
            self.yref[j] = np.concatenate((stacked[j, :], u_target[j, :]))
        self.yref_e[:] = stacked[self.N, :]
        return 0

    def run_optimization(self, initial_state=None, use_model=0, return_x=False, gp_regression_state=None):
        """:527-566: one RTI step from the stored iterate; returns the flattened input sequence (and the states)."""
        if initial_state is None:
            initial_state = [0, 0, 0] + [1, 0, 0, 0] + [0, 0, 0] + [0, 0, 0]
        x_init = np.stack(initial_state).astype(np.float64).reshape(1, QNX)
        x, u, cost, st, it = self.solver.solve_numpy(x_init, self.yref[None], self.yref_e[None], self.x_iter[None], self.u_iter[None])
        self.status = int(st[0])
        if self.status == 0:
            self.x_iter, self.u_iter = x[0], u[0]
        w_opt = np.reshape(self.u_iter.copy(), (-1))
        return w_opt if not return_x else (w_opt, self.x_iter.copy())
