"""ROS adapter with the interface of the reference's ``ROSGPMPC``
(data_driven_mpc/ros_gp_mpc/src/ad_mpc/create_ros_ad_mpc.py:41-101).

rospy / ackermann_msgs are used when importable; otherwise plain stand-in message classes with the
same field names (AckermannDrive = 5 x float32: steering_angle, steering_angle_velocity, speed,
acceleration, jerk -- msgs_pkg/ackermann_msgs/msg/AckermannDrive.msg) keep the return tuple intact.
"""
import time

import numpy as np

from . import config as _c
from .ad_3d import AD3D
from .ad_3d_mpc import AD3DMPC

try:  # pragma: no cover - ROS is not installed on the build / GPU boxes
    import rospy
    import std_msgs.msg
    from ackermann_msgs.msg import AckermannDrive, AckermannDriveStamped
    _Header = std_msgs.msg.Header
    _now = rospy.Time.now
except Exception:  # noqa: BLE001
    class _Header:
        def __init__(self):
            self.seq = 0
            self.stamp = 0.0
            self.frame_id = ""

    class AckermannDrive:
        __slots__ = ("steering_angle", "steering_angle_velocity", "speed", "acceleration", "jerk")

        def __init__(self):
            for f in self.__slots__:
                setattr(self, f, 0.0)

        def __setattr__(self, k, v):          # message fields are float32
            object.__setattr__(self, k, float(np.float32(v)))

    class AckermannDriveStamped:
        def __init__(self):
            self.header = _Header()
            self.drive = AckermannDrive()

    _now = time.time


def custom_ad_param_loader(ad_name):
    return AD3D(noisy=False, noisy_input=False)       # create_ros_ad_mpc.py:26-38


class ROSGPMPC:
    def __init__(self, t_horizon, n_mpc_nodes, opt_dt, ad_name="sim_car", point_reference=False, device=0):
        ad = custom_ad_param_loader(ad_name)
        if point_reference:                            # :46-56
            acados_config = {"solver_type": "SQP", "terminal_cost": True}
        else:
            acados_config = {"solver_type": "SQP_RTI", "terminal_cost": False}
        q_diagonal = np.array(_c.Q_DIAG_ROS)           # :58
        r_diagonal = np.array(_c.R_DIAG_ROS)           # :59
        self.ad_mpc = AD3DMPC(ad, t_horizon=t_horizon, optimization_dt=opt_dt, n_nodes=n_mpc_nodes, model_name=ad_name,
                              solver_options=acados_config, q_cost=q_diagonal, r_cost=r_diagonal, device=device)
        self.ad_name = ad_name
        self.ad = ad
        self.last_w = None

    def set_state(self, x):
        self.ad.set_state(x)

    def set_reference(self, x_ref, u_ref, terminal_point=False):
        return self.ad_mpc.set_reference(x_reference=x_ref, u_reference=u_ref, terminal_point=terminal_point)

    def optimize(self, model_data):
        """-> (AckermannDriveStamped, w_opt, x_opt, solver_status)  (create_ros_ad_mpc.py:88-101)."""
        w_opt, x_opt, solver_status = self.ad_mpc.optimize(use_model=model_data, return_x=True)
        msg = AckermannDriveStamped()
        msg.header = _Header()
        msg.header.stamp = _now()
        msg.drive.steering_angle = x_opt[0, 6]
        msg.drive.steering_angle_velocity = w_opt[1]
        msg.drive.speed = x_opt[0, 3]
        msg.drive.acceleration = w_opt[0]
        return msg, w_opt, x_opt, solver_status
