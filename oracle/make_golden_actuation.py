"""Writes tests/golden/actuation.json: small cases of resample_vel and of the post-solve branch whose expected values follow from
the reference lines by hand (each case says how).  Plain Python arithmetic, no solver involved."""
import json, math, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
f32 = lambda v: float(np.float32(v))
cases = {"resample_vel": [], "actuation": []}
# --- resample_vel (gp_ad_mpc_node.py:344-349)
# |v| = 5 exactly (3-4-5), increment 5 * 0.1 * 0.8 = 0.4 by repeated addition
b = 5.0; exp = []
for i in range(6):
    exp.append(b); b = b + 5 * 0.1 * 0.8
cases["resample_vel"].append(dict(why="every reference above the bound: output is the bound sequence 5, 5.4, 5.8, ... (repeated addition)",
                                  vx=3.0, vy=4.0, acc_max=5.0, dt=0.1, vel_ref=[10.0] * 6, expected=exp))
cases["resample_vel"].append(dict(why="references below the bound are untouched", vx=3.0, vy=4.0, acc_max=5.0, dt=0.1,
                                  vel_ref=[1.0, 2.0, 3.0, 4.0], expected=[1.0, 2.0, 3.0, 4.0]))
cases["resample_vel"].append(dict(why="mixed: slot 0 clamped to |v| = 2, slot 1 (2.1) below 2 + 0.2, slot 2 (9) clamped to 2 + 0.2 + 0.2",
                                  vx=2.0, vy=0.0, acc_max=5.0, dt=0.05, vel_ref=[3.0, 2.1, 9.0],
                                  expected=[2.0, 2.1, (2.0 + 5.0 * 0.05 * 0.8) + 5.0 * 0.05 * 0.8]))
# --- actuation: straight reference along x, prediction offset laterally by `off` (distance = |off| on every slot but the last)
def traj(N, off, delta0=0.1, v0=5.0):
    ref = np.zeros((N + 1, 7)); ref[:, 0] = np.arange(N + 1) * 0.25
    x = ref.copy(); x[:, 1] = off; x[0, 6] = delta0; x[0, 3] = v0
    return x, ref
N = 4
def add(why, status, off, w, steering, cnt, thr, exp_cnt, exp_mode, exp_rec, exp_healthy):
    x, ref = traj(N, off)
    cases["actuation"].append(dict(why=why, N=N, status=status, x_opt=x.tolist(), w_opt=w, ref=ref.tolist(), steering=steering, safe_count=cnt,
                                   threshold=thr, expected=dict(safe_count=exp_cnt, mode=exp_mode, record=list(exp_rec), healthy=exp_healthy)))
w = [0.5, 1.0] + [0.0] * (2 * N - 2)
add("9 earlier successes + this one = 10 -> MPC command; steering = clip(1.0) * 0.1 + 0.2 = 0.30000000000000004 -> float32", 0, 0.5, w, 0.2, 9, 10,
    10, 1, (f32(1.0 * 0.1 + 0.2), 1.0, 5.0, 0.5), True)
add("only 9 successes: auxiliary controller (hold steering 0.2, brake -1e5)", 0, 0.5, w, 0.2, 8, 10, 9, 0, (f32(0.2), 0.0, 0.0, f32(-1e5)), True)
add("solver status 4 resets the counter: auxiliary controller", 4, 0.5, w, 0.2, 50, 10, 0, 0, (f32(0.2), 0.0, 0.0, f32(-1e5)), True)
add("prediction 3.9 m off on 4 of 5 slots: mean 3.12 >= 3 -> unhealthy, auxiliary controller although the counter passes", 0, 3.9, w, -0.1, 20, 10,
    21, 0, (f32(-0.1), 0.0, 0.0, f32(-1e5)), False)
w2 = [-2.0, 7.5] + [0.0] * (2 * N - 2)
add("steering rate 7.5 clipped to 3 -> 3 * 0.1 + 0.4 = 0.7 clipped to the steering bound 0.52; the message keeps the raw rate 7.5", 0, 0.0, w2, 0.4, 10, 10,
    11, 1, (f32(0.52), 7.5, 5.0, -2.0), True)
w3 = [1.25, -9.0] + [0.0] * (2 * N - 2)
add("rate -9 clipped to -3 -> -0.3 - 0.3 = -0.6 clipped to -0.52", 0, 0.0, w3, -0.3, 10, 10, 11, 1, (f32(-0.52), -9.0, 5.0, 1.25), True)
json.dump(cases, open(os.path.join(HERE, "..", "tests", "golden", "actuation.json"), "w"), indent=1)
print("wrote", len(cases["resample_vel"]), "+", len(cases["actuation"]), "cases")
