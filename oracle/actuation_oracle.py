"""Loop-form numpy restatement of the node logic on either side of the solve.  TEST INFRASTRUCTURE ONLY.

Follows data_driven_mpc/ros_gp_mpc/nodes/gp_ad_mpc_node.py line by line:
  resample_vel      :344-349
  check_pred_trj    :248-257
  run_mpc tail      :199-235   (success counter, gate, steering post-processing; message fields are float32)
  run_pure          :455-476   (auxiliary controller: hold steering, acceleration -1e5)
and src/ad_mpc/create_ros_ad_mpc.py:92-98 (Ackermann mapping).  Pinned by tests/golden/actuation.json (hand-derived cases)."""
import math

import numpy as np


def resample_vel(vel_ref, v_x, v_y, acc_max, dt):
    vel_ref = [float(v) for v in vel_ref]
    # :345 reads math.sqrt(self.v_x**2 + self.v_y**2).  CPython evaluates x**2 through libm's pow(), whose result differs from the
    # exact product by one ulp on a machine-dependent share of inputs (0.03 % in the build container, 4 % on the GPU box's host
    # CPU: measured).  The restatement squares by multiplication -- the correctly rounded value of what the line means.
    MAX_bound = math.sqrt(v_x * v_x + v_y * v_y)
    for i in range(len(vel_ref)):                                   # :346
        if vel_ref[i] > MAX_bound:                                  # :347
            vel_ref[i] = MAX_bound                                  # :348
        MAX_bound = MAX_bound + acc_max * dt * 0.8                  # :349
    return vel_ref


def check_pred_trj(x_opt, ref):
    tmp_dist = np.zeros(len(ref))                                   # :250
    for i in range(0, len(ref) - 1):                                # :251
        tmp_dist[i] = math.sqrt((ref[i, 0] - x_opt[i, 0]) ** 2 + (ref[i, 1] - x_opt[i, 1]) ** 2)
    return bool(np.mean(tmp_dist) < 3.0 and np.cov(tmp_dist) < 2 and np.max(tmp_dist) < 4)      # :254


def actuation(solver_status, x_opt, w_opt, ref, steering, mpc_safe_count, threshold,
              steering_rate_min=-3.0, steering_rate_max=3.0, steering_min=-0.52, steering_max=0.52):
    """One pass of the node after optimize().  Returns (mpc_safe_count, mode, record, healthy); record = the published AckermannDrive
    fields (steering_angle, steering_angle_velocity, speed, acceleration) as float32 values, mode 1 = MPC command, 0 = auxiliary."""
    f32 = lambda v: float(np.float32(v))
    # create_ros_ad_mpc.py:95-98 -- float32 message fields
    drive = dict(steering_angle=f32(x_opt[0, 6]), steering_angle_velocity=f32(w_opt[1]), speed=f32(x_opt[0, 3]), acceleration=f32(w_opt[0]))
    healthy = check_pred_trj(x_opt, ref)                            # :203
    if solver_status > 0:                                           # :207
        mpc_safe_count = 0
    else:
        mpc_safe_count = mpc_safe_count + 1                         # :211
    publish = not (mpc_safe_count < threshold) and healthy          # :213-217
    if publish:
        steering_val = max(min(steering_rate_max, drive["steering_angle_velocity"]), steering_rate_min)       # :222
        drive["steering_angle"] = f32(max(min(steering_max, steering_val * 0.1 + steering), steering_min))    # :223
        return mpc_safe_count, 1, (drive["steering_angle"], drive["steering_angle_velocity"], drive["speed"], drive["acceleration"]), healthy
    # :455-476 run_pure: a fresh AckermannDrive() with the measured steering and a hard brake
    return mpc_safe_count, 0, (f32(steering), 0.0, 0.0, f32(-1e5)), healthy
