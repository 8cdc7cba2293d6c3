"""Host-side orchestration pieces of ``AD3DOptimizer.run_optimization`` (H7/H8 of SURVEY 8a),
vectorised over a leading batch axis.  Pure numpy; no solver arithmetic lives here.

Reference lines are relative to data_driven_mpc/ros_gp_mpc/src/ad_mpc/.
"""
import math

import numpy as np


def yaw_fix(psi0, psi_ref):
    """+-2*pi fix of the yaw reference relative to the sign of the initial yaw
    (ad_3d_optimizer.py:423-437): if psi0<0 and psi0+pi<ref -> ref-2pi; if psi0>0 and
    psi0-pi>ref -> ref+2pi; psi0==0 -> unchanged.  ``psi0`` broadcasts against ``psi_ref``."""
    psi0 = np.asarray(psi0, dtype=np.float64)
    ref = np.array(psi_ref, dtype=np.float64, copy=True)
    neg = (psi0 < 0) & (psi0 + math.pi < ref)
    pos = (psi0 > 0) & (psi0 - math.pi > ref)
    return np.where(neg, ref - 2 * math.pi, np.where(pos, ref + 2 * math.pi, ref))


def vel_switch(vx, blend_min, blend_max):
    """Blend parameter p = clip((vx-blend_min)/(blend_max-blend_min), 0, 1)  (ad_3d_optimizer.py:443)."""
    return np.minimum(np.maximum((np.asarray(vx, dtype=np.float64) - blend_min) / (blend_max - blend_min), 0.0), 1.0)


def pad_reference(x_target, u_target, N):
    """Repeat the last row until there are N+1 state rows (ad_3d_optimizer.py:347-349).  The reference
    appends one u row per appended x row, so u ends up with rows_u + (N+1-rows_x) rows."""
    x_target = np.array(x_target, dtype=np.float64, copy=True)
    u_target = np.array(u_target, dtype=np.float64, copy=True)
    while x_target.shape[0] < N + 1:
        x_target = np.vstack((x_target, x_target[-1, :]))
        u_target = np.vstack((u_target, u_target[-1, :]))
    return x_target, u_target


def is_valid_command(x_opt, ref):
    """Trajectory sanity test of ad_3d_optimizer.py:385-394: XY distance between prediction and
    reference over len(ref) slots of which the last one stays 0; mean<3, unbiased variance<2, max<4."""
    n = len(ref)
    d = np.zeros(n)
    d[: n - 1] = np.sqrt((ref[: n - 1, 0] - x_opt[: n - 1, 0]) ** 2 + (ref[: n - 1, 1] - x_opt[: n - 1, 1]) ** 2)
    return bool(np.mean(d) < 3.0 and np.cov(d) < 2 and np.max(d) < 4)


def fallback_command(prev_w):
    """The reference's off-by-one "shift" of the previous inputs (ad_3d_optimizer.py:475): yields 2N-1
    elements; reproduced verbatim because callers index w_opt[0], w_opt[1] only."""
    return np.concatenate((prev_w[2:-1], prev_w[-3:-1]))


def ackermann_fields(x_opt, w_opt):
    """create_ros_ad_mpc.py:95-98 -> (steering_angle, steering_angle_velocity, speed, acceleration)."""
    return float(x_opt[0, 6]), float(w_opt[1]), float(x_opt[0, 3]), float(w_opt[0])


def resample_vel(vel_ref, vx, vy, acc_max, dt):
    """gp_ad_mpc_node.py:344-349: acceleration-limited clamp of the speed reference (returns a new list)."""
    out = [float(v) for v in vel_ref]
    bound = math.sqrt(vx * vx + vy * vy)          # the reference writes **2 (libm pow: machine-dependent last bit); see oracle/actuation_oracle.py
    for i in range(len(out)):
        if out[i] > bound:
            out[i] = bound
        bound = bound + acc_max * dt * 0.8
    return out


def actuation(status, healthy, safe_count, threshold, ack_fields, steer_meas, rate_min, rate_max, steer_min, steer_max):
    """The node's branch after optimize() (gp_ad_mpc_node.py:199-235, :455-476) for one slot.
    ack_fields = ackermann_fields(...) as they sit in the float32 message.  Returns (safe_count, mode, record) with record =
    (steering_angle, steering_angle_velocity, speed, acceleration) as float32 values."""
    safe_count = 0 if status > 0 else safe_count + 1
    f32 = lambda v: float(np.float32(v))
    if safe_count >= threshold and healthy:
        rate = f32(ack_fields[1])
        sv = max(min(rate_max, rate), rate_min)
        ang = max(min(steer_max, sv * 0.1 + steer_meas), steer_min)
        return safe_count, 1, (f32(ang), rate, f32(ack_fields[2]), f32(ack_fields[3]))
    return safe_count, 0, (f32(steer_meas), 0.0, 0.0, f32(-1e5))
