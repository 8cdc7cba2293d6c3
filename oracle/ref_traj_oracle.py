"""CPU restatement (numpy, loop form) of RefTrajectory.get_waypoints -- TEST INFRASTRUCTURE ONLY.

Follows data_driven_mpc/ros_gp_mpc/src/ad_mpc/ref_traj.py:89-171 statement by statement on a trajectory table
[vel, x, y, psi, cdist, curv] (built as in :67-86).  Pinned by tests/golden/ref_traj.json, whose vectors were produced by
importing the reference module itself (oracle/make_golden.py)."""
import numpy as np


def bound(a):
    return (a + np.pi) % (2.0 * np.pi) - np.pi                      # :29-30


def fix_angle_reference(angle_ref, angle_init):                     # :32-37
    diff = bound(angle_ref - angle_init)
    diff = np.unwrap(diff)
    return angle_init + diff


def get_waypoints(traj, H, dt, X_init, Y_init, psi_init):
    vel, x, y, psi, cdist, curv = (traj[:, i] for i in range(6))
    psi_init = bound(psi_init)                                      # :94
    d = np.sqrt((x - X_init) ** 2 + (y - Y_init) ** 2)              # :101 (norm over axis 1)
    ci = int(np.argmin(d))
    pw = psi[ci]
    ex, ey = X_init - x[ci], Y_init - y[ci]
    out = {"s0": cdist[ci], "e_y0": -np.sin(pw) * ex + np.cos(pw) * ey, "e_psi0": bound(psi_init - pw)}     # :108-116
    v = list(vel)
    while len(v) < H + 1:                                           # :127-128
        v.append(0.01)
    s = [dt * v[0]]
    for h in range(1, H):                                           # :130-132
        s.append(s[-1] + dt * v[h])
    out["x_ref"] = np.interp(s, cdist, x); out["y_ref"] = np.interp(s, cdist, y)
    out["cdist_ref"] = np.interp(s, cdist, cdist); out["curv_ref"] = np.interp(s, cdist, curv)
    p = np.interp(s, cdist, np.unwrap(psi))                         # :137-140
    out["psi_ref"] = bound(fix_angle_reference(p, psi_init))        # :146-148
    vr = np.diff(out["cdist_ref"]) / dt                             # :151-152
    out["v_ref"] = np.insert(vr, len(vr), vr[-1])
    out["stop"] = bool(out["cdist_ref"][-1] == cdist[-1])           # :154-156
    out["x_ref"] = np.hstack([np.linspace(X_init, out["x_ref"][1], 3), out["x_ref"][2:-1]])      # :160-161
    out["y_ref"] = np.hstack([np.linspace(Y_init, out["y_ref"][1], 3), out["y_ref"][2:-1]])
    out["psi_ref"] = np.hstack([np.ones(3) * out["psi_ref"][0], out["psi_ref"][2:-1]])
    out["v_ref"] = np.hstack([np.ones(3) * out["v_ref"][2], out["v_ref"][2:-1]])
    return out
