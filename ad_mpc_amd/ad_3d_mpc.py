"""Facade with the interface of the reference's ``AD3DMPC``
(data_driven_mpc/ros_gp_mpc/src/ad_mpc/ad_3d_mpc.py:22-110)."""
import numpy as np

from .ad_3d_optimizer import AD3DOptimizer


class AD3DMPC:
    def __init__(self, my_ad, t_horizon=1.0, n_nodes=20, q_cost=None, r_cost=None, optimization_dt=5e-2,
                 simulation_dt=5e-4, model_name="my_ad", solver_options=None, device=0):
        self.ad = my_ad
        self.simulation_dt = simulation_dt
        self.optimization_dt = optimization_dt
        self.pre_u = np.array([0.0, 0.0])
        self.n_nodes = n_nodes
        self.t_horizon = t_horizon
        self.ad_opt = AD3DOptimizer(my_ad, t_horizon=t_horizon, n_nodes=n_nodes, q_cost=q_cost, r_cost=r_cost,
                                    model_name=model_name, solver_options=solver_options, device=device)

    def clear(self):
        self.ad_opt.clear_acados_model()

    def get_state(self):
        """7x1 column of the current vehicle state (ad_3d_mpc.py:54-60)."""
        return np.expand_dims(self.ad.get_state(stacked=True), 1)

    def set_reference(self, x_reference, u_reference=None, terminal_point=False):
        """Point target if x_reference[0] is a list or terminal_point, else a trajectory (ad_3d_mpc.py:62-76)."""
        if isinstance(x_reference[0], list) or terminal_point:
            return self.ad_opt.set_reference_state(x_reference, u_reference)
        return self.ad_opt.set_reference_trajectory(x_reference, u_reference)

    def optimize(self, use_model=0, return_x=False):
        """ad_3d_mpc.py:78-93: w_opt[2N], or (w_opt, x_opt[N+1,7], status) with return_x."""
        state = np.expand_dims(self.ad.get_state(stacked=True), 0)       # shape (1,7), :89
        return self.ad_opt.run_optimization(state, use_model=use_model, return_x=return_x)

    @staticmethod
    def reshape_input_sequence(u_seq):
        """Kept for interface parity (ad_3d_mpc.py:96-107; a quadrotor left-over that regroups by 4)."""
        k = np.arange(u_seq.shape[0] / 4, dtype=int)
        u_seq = np.atleast_2d(u_seq).T if len(u_seq.shape) == 1 else u_seq
        return np.concatenate((u_seq[4 * k], u_seq[4 * k + 1], u_seq[4 * k + 2], u_seq[4 * k + 3]), 1)

    def reset(self):
        return
