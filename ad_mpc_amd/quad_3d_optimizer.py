"""Host mirror of the reference's ``Quad3DOptimizer`` (src/quad_mpc/quad_3d_optimizer.py) over the quadrotor path of the engine
(include/admpc_quad.h): same constructor arguments that shape the OCP, ``set_reference_state`` / ``set_reference_trajectory`` /
``run_optimization`` with the reference's quirks (weight expansion for the quaternion, body-frame velocity in the point reference,
last-row padding of short trajectories, persistent never-shifted iterate).  One vehicle = a batch of one."""
from copy import copy

import numpy as np
import torch

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


class QuadGPEnsemble:
    """What the optimiser needs of the reference's ``GPEnsemble`` (src/model_fitting/gp.py) for the quadrotor: per cluster the regressors
    of the body-frame acceleration components (``set_quad_gp`` dicts), the K x d centroids and the features they live in (indices into
    z = [x with the velocity in the body frame; u]).  Homogeneous ensembles only -- the reference's MPC path keys its solvers with ONE
    index per solve (quad_3d_optimizer.py:452-462)."""

    def __init__(self, clusters, centroids, feats):
        self.clusters = [list(c) for c in clusters]
        self.centroids = np.asarray(centroids, dtype=np.float64).reshape(len(self.clusters), -1)
        self.feats = [int(f) for f in np.atleast_1d(feats).reshape(-1)]
        self.n_models = len(self.clusters)

    def select_gp(self, z):
        """gp.py:738-770: nearest centroid of the selected features; z = the full 17-vector [x (body-frame velocity); u]."""
        zz = np.asarray(z, dtype=np.float64)[self.feats]
        return int(np.argmin(np.sqrt(((zz[None, :] - self.centroids) ** 2).sum(1))))


class Quad3DOptimizer:
    def __init__(self, quad=None, t_horizon=1, n_nodes=20, q_cost=None, r_cost=None, q_mask=None,
                 B_x=None, gp_regressors=None, rdrv_d_mat=None, model_name="quad_3d_acados_mpc", solver_options=None, device=0):
        """quad_3d_optimizer.py:29-207 (same argument order).  ``quad``: an object with mass, J, max_thrust, x_f, y_f, z_l_tau, min_u /
        max_u (or max_input_value / min_input_value) as ``Quadrotor3D`` has them; None = the vehicle of quad_3d.py.
        ``gp_regressors``: a ``QuadGPEnsemble`` (or a plain list of ``set_quad_gp`` dicts = one cluster) -- one engine handle per cluster
        as the reference keeps one acados solver per cluster (:207), each with its own persistent iterate.  ``B_x`` is implied by the
        regressors' ``out`` entries (7, 8, 9: the velocity rows) and only accepted for signature compatibility.  ``rdrv_d_mat``: the 3 x 3
        (diagonal) linear drag matrix of :364-381."""
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
        # :203  nlp_solver_type = 'SQP_RTI' if solver_options is None else solver_options["solver_type"]; create_ros_gp_mpc.py:63-68 asks for
        # "SQP" with point references.  "SQP": acados' loop -- at most nlp_solver_max_iter = 100 QPs, stopping on the four KKT residual norms
        # <= 1e-6 (my_quad_acados_ocp.json:2075-2080), status 2 at the limit.  Anything else is refused, never ignored.
        solver_type = "SQP_RTI" if solver_options is None else solver_options.get("solver_type", "SQP_RTI")
        if solver_type == "SQP_RTI":
            cfg.sqp_iters, cfg.sqp_tol = 1, 0.0
        elif solver_type == "SQP":
            cfg.sqp_iters, cfg.sqp_tol = 100, 1e-6
            if gp_regressors is not None and isinstance(gp_regressors, QuadGPEnsemble) and len(gp_regressors.clusters) > 1:
                raise NotImplementedError('solver_type "SQP" with a clustered GP ensemble (routed solves) is not implemented')
        else:
            raise ValueError('solver_options["solver_type"] must be "SQP_RTI" or "SQP" (quad_3d_optimizer.py:203), got %r' % (solver_type,))
        self.solver_type = solver_type
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
        if rdrv_d_mat is not None:
            d = np.asarray(rdrv_d_mat, dtype=np.float64)
            if d.shape != (3, 3) or np.abs(d - np.diag(np.diag(d))).max() != 0.0:
                raise ValueError("rdrv_d_mat must be a 3 x 3 diagonal matrix (quad_3d_optimizer.py:364-381)")
            for i in range(3):
                cfg.rdrv[i] = float(d[i, i])
        self.cfg = cfg
        if gp_regressors is not None and not isinstance(gp_regressors, QuadGPEnsemble):
            gp_regressors = QuadGPEnsemble([list(gp_regressors)], np.zeros((1, 1)), [7])
        self.gp_reg_ensemble = gp_regressors
        self.with_gp = gp_regressors is not None
        if self.with_gp:
            from .quad_config import set_quad_gp
            self.solvers = []
            for gps in gp_regressors.clusters:
                cc = cfg.copy(); set_quad_gp(cc, gps)
                self.solvers.append(QuadBatchSolver(cc, device=device))
        else:
            self.solvers = [QuadBatchSolver(cfg, device=device)]
        self.solver = self.solvers[0]
        K = len(self.solvers)
        # every acados solver of the reference has its own references and its own (zero-initialised, never shifted) iterate
        self.yref = [np.zeros((self.N, QNX + QNU)) for _ in range(K)]; self.yref_e = [np.zeros(QNX) for _ in range(K)]
        self.x_iter = [np.zeros((self.N + 1, QNX)) for _ in range(K)]; self.u_iter = [np.zeros((self.N, QNU)) for _ in range(K)]
        self.target = None
        self.status = 0
        from .engine import PackedIO
        from .quad_config import QNY
        self._io = PackedIO(self.solver.device, (("x0", (1, QNX)), ("yref", (1, self.N, QNY)), ("yref_e", (1, QNX)), ("gp", (1, QNX)),
                                                 ("x", (1, self.N + 1, QNX)), ("u", (1, self.N, QNU))), (("cost", (1,)),))

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
        gp_ind = self.gp_reg_ensemble.select_gp(ref) if self.with_gp else 0          # :449-452: from the reference, velocity in the body frame
        self.yref[gp_ind][:] = ref
        self.yref_e[gp_ind][:] = ref[:-4]
        return gp_ind

    def set_reference_trajectory(self, x_target, u_target):
        """:465-503: short trajectories are padded with their last row; the last node has a state reference only."""
        if u_target is not None:
            assert x_target[0].shape[0] == (u_target.shape[0] + 1) or x_target[0].shape[0] == u_target.shape[0]
        while x_target[0].shape[0] < self.N + 1:
            x_target = [np.concatenate((x, np.expand_dims(x[-1, :], 0)), 0) for x in x_target]
            if u_target is not None:
                u_target = np.concatenate((u_target, np.expand_dims(u_target[-1, :], 0)), 0)
        stacked = np.concatenate([x for x in x_target], 1)
        if self.with_gp:                                          # :484-491: the model is chosen from the middle of the reference
            x_mean = stacked[int(self.N / 2)]
            v_b = v_dot_q(x_mean[7:10], quaternion_inverse(x_mean[3:7]))
            gp_ind = self.gp_reg_ensemble.select_gp(np.concatenate((x_mean[:7], v_b, x_mean[10:], u_target[int(self.N / 2)])))
        else:
            gp_ind = 0
        self.target = copy(x_target)
        for j in range(self.N):
            self.yref[gp_ind][j] = np.concatenate((stacked[j, :], u_target[j, :]))
        self.yref_e[gp_ind][:] = stacked[self.N, :]
        return gp_ind

    def run_optimization(self, initial_state=None, use_model=0, return_x=False, gp_regression_state=None):
        """:527-566: one RTI step from the stored iterate; returns the flattened input sequence (and the states)."""
        if initial_state is None:
            initial_state = [0, 0, 0] + [1, 0, 0, 0] + [0, 0, 0] + [0, 0, 0]
        x_init = np.stack(initial_state).astype(np.float64).reshape(1, QNX)
        m = int(use_model)
        gp_state = None
        if self.with_gp and gp_regression_state is not None:      # :546-552: p = [gp_state, 1] at node 0 (default: the initial state)
            gp_state = np.asarray(gp_regression_state, dtype=np.float64).reshape(1, QNX)
        io, sv = self._io, self.solvers[m]
        io.put("x0", x_init); io.put("yref", self.yref[m]); io.put("yref_e", self.yref_e[m]); io.put("gp", x_init if gp_state is None else gp_state)
        io.put("x", self.x_iter[m]); io.put("u", self.u_iter[m])
        with torch.cuda.device(io.device):
            io.upload()
            sv.solve(io.dev("x0"), io.dev("yref"), io.dev("yref_e"), io.dev("x"), io.dev("u"), io.dev("cost"), io.dint[0:1], io.dint[1:2],
                     gp_state=None if gp_state is None else io.dev("gp"))
            io.download("x")
        self.status = int(io.hi[0])
        if self.status in (0, 2):                 # 2: solver_type "SQP" at its iteration limit -- acados keeps (and so returns) the last iterate
            self.x_iter[m], self.u_iter[m] = io.take("x")[0], io.take("u")[0]
        w_opt = np.reshape(self.u_iter[m].copy(), (-1))
        return w_opt if not return_x else (w_opt, self.x_iter[m].copy())
