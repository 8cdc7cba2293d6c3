#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.  TEST INFRASTRUCTURE ONLY.

Runs only in the development container (needs /root/reference and oracle/_ref built by
`make -C oracle ref`).  The fixtures are DATA (inputs and expected outputs); no reference source
text is copied.

  tests/golden/shooting.json
      120 random (x, u, p) points -> xdot, phi, A, B computed by the REFERENCE's own compiled
      CasADi model code (c_generated_code/sim_car_model/sim_car_expl_ode_fun.c and
      sim_car_expl_vde_forw.c) driven as acados' ERK4 does (acados_solver_sim_car.c:655-665).

  tests/golden/kat_sim_car_iterate.json
      Known-answer test for the solver, reconstructed from the converged acados iterate that the
      reference ships (src/ad_mpc/sim_car_iterate.json, N=40, kinematic mode p=0):
      x0, the stage references recovered from the stored multipliers through the stationarity
      conditions, the terminal weight scale that makes the fixture
      consistent (1e-2, not the shipped 1e-6), and the stored X, U, pi, lam themselves.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:] = [q for q in sys.path if os.path.abspath(q or '.') != HERE]   # `oracle` must resolve to the package
sys.path.insert(0, ROOT)

from oracle.oracle import RefModel  # noqa: E402

REF = os.environ.get("ADMPC_REFERENCE", "/root/reference")
ITERATE = os.path.join(REF, "data_driven_mpc/ros_gp_mpc/src/ad_mpc/sim_car_iterate.json")
OUT = os.path.join(ROOT, "tests", "golden")
NX, NU = 7, 2
TS = 0.05


def shooting(ref):
    rng = np.random.default_rng(20261003)
    cases = []
    for i in range(120):
        p = [0.0, 0.3, 1.0][i % 3]
        x = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-np.pi, np.pi), rng.uniform(0.5, 20),
                      rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.5, 0.5)])
        u = np.array([rng.uniform(-10, 5), rng.uniform(-3, 3)])
        if i >= 114:      # the reference's own initial iterate: all zeros (acados_solver_sim_car.c:705-731)
            x = np.zeros(NX); u = np.zeros(NU); p = 0.0
            if i >= 117:
                x[0:3] = rng.uniform(-3, 3, 3); u = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1)])
        xdot = ref.ode(x, u, p)
        phi, A, B = ref.rk4_sens(x, u, p, TS)
        cases.append(dict(x=x.tolist(), u=u.tolist(), p=p, h=TS, xdot=xdot.tolist(), phi=phi.tolist(),
                          A=A.tolist(), B=B.tolist()))
    return dict(source="reference CasADi C (sim_car_expl_ode_fun.c, sim_car_expl_vde_forw.c) via oracle/_ref",
                cases=cases)


def kat(ref):
    d = json.load(open(ITERATE))
    N = 40
    X = np.array([d["x_%d" % k] for k in range(N + 1)])
    U = np.array([d["u_%d" % k] for k in range(N)])
    PI = np.array([d["pi_%d" % k] for k in range(N)])
    LAM = [np.array(d["lam_%d" % k]) for k in range(N)]
    q = np.array([10.0, 10.0, 100.0, 0, 0, 0, 0]); r = np.array([1.0, 100.0])
    gaps = 0.0
    A = np.zeros((N, NX, NX)); B = np.zeros((N, NX, NU))
    for k in range(N):
        phi, A[k], B[k] = ref.rk4_sens(X[k], U[k], 0.0, TS)
        gaps = max(gaps, np.abs(phi - X[k + 1]).max())
    # multiplier layout per stage (SURVEY 8c Pin 2 iii): k>=1: [lbu(2), lbx(1), ubu(2), ubx(1), ls(2), us(2)]
    #                                                    k==0: [lbu(2), lbx(7), ubu(2), ubx(7), ls(2), us(2)]
    yref = np.zeros((N, NX + NU))
    uref_rec = np.zeros((N, NU))
    for k in range(N):
        lam = LAM[k]
        nbx = 7 if k == 0 else 1
        lbu = lam[0:2]; lbx = lam[2:2 + nbx]; ubu = lam[2 + nbx:4 + nbx]; ubx = lam[4 + nbx:4 + 2 * nbx]
        # stationarity in u:  Ts*R*(u-uref) + B'pi - lam_lbu + lam_ubu = 0
        uref_rec[k] = U[k] + (B[k].T @ PI[k] - lbu + ubu) / (TS * r)
        if k >= 1:
            g = A[k].T @ PI[k] - PI[k - 1]
            g[6] += -lbx[0] + ubx[0]
            # stationarity in x: Ts*Q*(x-xref) + A'pi_k - pi_{k-1} - lam_lbx + lam_ubx = 0 (weighted rows only)
            yref[k, 0:3] = X[k, 0:3] + g[0:3] / (TS * q[0:3])
            res_unweighted = np.abs(g[3:]).max()
            assert res_unweighted < 1e-6, res_unweighted
        else:
            yref[0, 0:3] = X[0, 0:3]     # stage-0 state reference is irrelevant (x_0 is fixed); use x0
    assert np.abs(uref_rec).max() < 1e-8, np.abs(uref_rec).max()
    yref[:, 3] = 0.0                     # weight 0 -> value irrelevant
    # terminal: pi_{N-1} = W_e (x_N - yref_N)  with W_e = 1e-2*q (consistent with last-row padding)
    we_scale = 1e-2
    We = we_scale * q
    yref_e = np.zeros(NX)
    yref_e[0:3] = X[N, 0:3] - PI[N - 1, 0:3] / We[0:3]
    pad_err = np.abs(yref_e[0:2] - yref[N - 1, 0:2]).max()
    return dict(
        source="data_driven_mpc/ros_gp_mpc/src/ad_mpc/sim_car_iterate.json (converged acados iterate, N=40, p=0)",
        N=N, Ts=TS, p=0.0, q=q.tolist(), r=r.tolist(), terminal_scale=we_scale,
        x0=X[0].tolist(), yref=yref.tolist(), yref_e=yref_e.tolist(),
        X=X.tolist(), U=U.tolist(), PI=PI.tolist(), LAM=[l.tolist() for l in LAM],
        checks=dict(max_shooting_gap=gaps, max_abs_recovered_uref=float(np.abs(uref_rec).max()),
                    terminal_xy_vs_last_row_padding=float(pad_err)))


ITERATE2 = os.path.join(REF, "data_driven_mpc/ros_gp_mpc/src/ad_mpc/solve_iteration.json")


def kat_mid_rti(ref):
    """Second, weaker pin from the reference's other stored iterate (src/ad_mpc/solve_iteration.json): N = 40, v_x 10 -> 14 m/s, taken in
    the middle of an RTI sequence with the DYNAMIC bicycle branch active (its shooting gaps are 8e-4 with p = 1, 0.077 with p = 0).
    What can be recovered exactly and what cannot:
      * x, y references: exactly.  Columns 0, 1 of every A_k are unit vectors whatever the linearisation point, so the x, y rows of the
        stationarity condition read Ts q (x_k - xref_k) + pi_k - pi_{k-1} = 0 with the stored quantities alone (the terminal point
        again equals the last row, to 7 digits, with the 1e-2 terminal scale -- the same consistency check as for the first fixture);
      * psi references: only to ~1e-3 rad.  Their row carries d(phi_x, phi_y)/d(psi) of the linearisation the QP was built at, i.e. at
        the PREVIOUS iterate, which the file does not hold; here they are evaluated at the stored iterate;
      * the iterate is one RTI step short of convergence: one step of a consistent solver started at it moves it by O(1e-2).
    Hence the test statement: started at the stored iterate with these references and p = 1, one RTI step stays within 2.5e-2 of it,
    so does the converged SQP solution, and the set of inputs at their bound (acceleration = 5 on stages 0..7) is the stored one."""
    d = json.load(open(ITERATE2))
    N = 40
    X = np.array([d["x_%d" % k] for k in range(N + 1)])
    U = np.array([d["u_%d" % k] for k in range(N)])
    PI = np.array([d["pi_%d" % k] for k in range(N)])
    LAM = [np.array(d["lam_%d" % k]) for k in range(N)]
    T = [np.array(d["t_%d" % k]) for k in range(N)]
    q = np.array([10.0, 10.0, 100.0, 0, 0, 0, 0])
    gap1 = gap0 = 0.0
    A = np.zeros((N, NX, NX)); B = np.zeros((N, NX, NU))
    for k in range(N):
        phi, A[k], B[k] = ref.rk4_sens(X[k], U[k], 1.0, TS)
        gap1 = max(gap1, np.abs(phi - X[k + 1]).max())
        gap0 = max(gap0, np.abs(ref.rk4_sens(X[k], U[k], 0.0, TS)[0] - X[k + 1]).max())
    yref = np.zeros((N, NX + NU))
    yref[0, 0:3] = X[0, 0:3]
    for k in range(1, N):
        lam = LAM[k]
        g = A[k].T @ PI[k] - PI[k - 1]
        g[6] += -lam[2] + lam[5]
        yref[k, 0:3] = X[k, 0:3] + g[0:3] / (TS * q[0:3])
    we_scale = 1e-2
    yref_e = np.zeros(NX)
    yref_e[0:3] = X[N, 0:3] - PI[N - 1, 0:3] / (we_scale * q[0:3])
    return dict(
        source="data_driven_mpc/ros_gp_mpc/src/ad_mpc/solve_iteration.json (acados iterate in the middle of an RTI sequence, N=40, dynamic branch)",
        N=N, Ts=TS, p=1.0, terminal_scale=we_scale, x0=X[0].tolist(), yref=yref.tolist(), yref_e=yref_e.tolist(),
        X=X.tolist(), U=U.tolist(), LAM=[l.tolist() for l in LAM],
        checks=dict(max_shooting_gap_p1=gap1, max_shooting_gap_p0=gap0,
                    terminal_xy_vs_last_row_padding=float(np.abs(yref_e[0:2] - yref[N - 1, 0:2]).max()),
                    max_lam_t=float(max((l * t).max() for l, t in zip(LAM, T)))))


def ref_traj_golden():
    """Golden vectors of RefTrajectory.set_traj/get_waypoints produced by the reference module itself.  ref_traj.py imports
    rosbag and rospy at module level without using them in these two functions; empty placeholder modules let it import."""
    import importlib.util
    import types
    for name in ("rosbag", "rospy"):
        sys.modules.setdefault(name, types.ModuleType(name))
    path = os.path.join(REF, "data_driven_mpc/ros_gp_mpc/src/ad_mpc/ref_traj.py")
    spec = importlib.util.spec_from_file_location("reference_ref_traj", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(7)
    cases = []
    for M, H, dt in [(400, 20, 0.05), (400, 40, 0.05), (60, 10, 0.2), (40, 45, 0.2)]:
        # a smooth path: varying curvature, yaw crossing +-pi, speeds 2..12 m/s
        s = np.linspace(0.0, 0.5 * M, M)
        psi = 2.8 + 0.9 * np.sin(s / 35.0) + 0.004 * s
        x = np.concatenate(([0.0], np.cumsum(np.cos(psi[:-1]) * np.diff(s)))) + 10.0
        y = np.concatenate(([0.0], np.cumsum(np.sin(psi[:-1]) * np.diff(s)))) - 5.0
        psi_w = (psi + np.pi) % (2 * np.pi) - np.pi
        vel = 7.0 + 5.0 * np.sin(s / 20.0)
        rt = mod.RefTrajectory(traj_horizon=H, traj_dt=dt)
        rt.set_traj(x, y, psi_w, vel)
        poses = []
        for _ in range(6):
            i = int(rng.integers(0, M))
            X0 = x[i] + rng.normal(0, 0.8); Y0 = y[i] + rng.normal(0, 0.8); P0 = psi_w[i] + rng.normal(0, 0.3) + rng.choice([0, 2 * np.pi, -2 * np.pi])
            w = rt.get_waypoints(X0, Y0, P0)
            poses.append(dict(X=X0, Y=Y0, psi=P0, out={k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in w.items()}))
        cases.append(dict(M=M, H=H, dt=dt, x=x.tolist(), y=y.tolist(), psi=psi_w.tolist(), vel=vel.tolist(),
                          table=rt.trajectory.tolist(), poses=poses))
    return dict(source="data_driven_mpc/ros_gp_mpc/src/ad_mpc/ref_traj.py imported in the development container", cases=cases)


def main():
    ref = RefModel()
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "shooting.json"), "w") as f:
        json.dump(shooting(ref), f)
    k = kat(ref)
    with open(os.path.join(OUT, "kat_sim_car_iterate.json"), "w") as f:
        json.dump(k, f)
    k2 = kat_mid_rti(ref)
    with open(os.path.join(OUT, "kat_solve_iteration.json"), "w") as f:
        json.dump(k2, f)
    with open(os.path.join(OUT, "ref_traj.json"), "w") as f:
        json.dump(ref_traj_golden(), f)
    print("shooting.json: 120 cases;  KAT checks:", k["checks"], "; mid-RTI fixture checks:", k2["checks"], "; ref_traj.json written")


if __name__ == "__main__":
    main()
