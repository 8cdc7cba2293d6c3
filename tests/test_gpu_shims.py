"""The reference's Python surface (AD3DMPC.set_reference/optimize, ROSGPMPC.optimize, the AcadosOcpSolver-shaped
seam) running on the HIP engine; BASELINE configs[0] (single vehicle, straight path) and the quirks of SURVEY 8b."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from ad_mpc_amd.config import default_config, Q_DIAG_ROS, R_DIAG_ROS  # noqa: E402
from ad_mpc_amd.scenarios import straight_scenario, assemble  # noqa: E402


def _ros_cfg(N=20, T=1.0):
    return default_config(N=N, Ts=T / N, q=Q_DIAG_ROS, r=R_DIAG_ROS)


def test_config1_through_the_ros_surface(oracle):
    from ad_mpc_amd.create_ros_ad_mpc import ROSGPMPC
    mpc = ROSGPMPC(t_horizon=1.0, n_mpc_nodes=20, opt_dt=0.01)
    x0, xref, uref = straight_scenario(N=20, Ts=0.05, v=5.0)
    x0 = x0.copy(); x0[1] = 0.4; x0[2] = 0.05                       # small lateral / heading error
    mpc.set_state(list(x0))
    # the node passes N reference rows and N-1 input rows (gp_ad_mpc_node.py:180-187); padding happens inside
    mpc.set_reference(xref[:20].copy(), uref[:19].copy())
    msg, w_opt, x_opt, status = mpc.optimize(0)
    assert status == 0 and w_opt.shape == (40,) and x_opt.shape == (21, 7)
    # oracle on the same inputs: padded reference (last row repeated), zero initial iterate, p = 0
    xr = np.vstack([xref[:20], xref[19:20]]); ur = np.vstack([uref[:19], uref[18:19]])
    s = assemble(x0[None], xr[None], ur[None], init="zeros")
    xo, uo, co, so, io = oracle.solve_batch(_ros_cfg(), s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    assert np.abs(w_opt - uo[0].reshape(-1)).max() < 1e-8 and np.abs(x_opt - xo[0]).max() < 1e-8
    d = msg.drive
    assert d.steering_angle == np.float32(x_opt[0, 6]) and d.steering_angle_velocity == np.float32(w_opt[1])
    assert d.speed == np.float32(x_opt[0, 3]) and d.acceleration == np.float32(w_opt[0])
    # second call: the solver keeps its iterate (never shifted, never reset) -> equals a second oracle step
    msg2, w2, x2, st2 = mpc.optimize(0)
    xo2, uo2, *_ = oracle.solve_batch(_ros_cfg(), s["x0"], s["yref"], s["yref_e"], s["p"], xo, uo)
    assert np.abs(w2 - uo2[0].reshape(-1)).max() < 1e-8


def test_optimizer_defaults_padding_and_fallback(oracle):
    from ad_mpc_amd.ad_3d import AD3D
    from ad_mpc_amd.ad_3d_mpc import AD3DMPC
    ad = AD3D()
    mpc = AD3DMPC(ad)                                     # optimizer-level default weights q=[10,10,50,0,0,0,1]
    assert list(mpc.ad_opt.ocp_config.W[:]) == [10, 10, 50, 0, 0, 0, 1, 1, 100]
    x0, xref, uref = straight_scenario()
    ad.set_state(list(x0))
    mpc.set_reference(xref[:5].copy(), uref[:4].copy())   # short reference -> padded to N+1 / N+... rows
    assert mpc.ad_opt.target.shape == (21, 7)
    w = mpc.optimize()
    assert w.shape == (40,) and np.isfinite(w).all() and w[0] < -1.0    # the padded (stopping) reference makes the MPC brake
    # make the prediction invalid (reference 50 m away): falls back to the previous inputs, 2N-1 long
    far = xref.copy(); far[:, 1] += 50.0
    mpc.set_reference(far, uref.copy())
    w2, x2, st = mpc.optimize(return_x=True)
    assert w2.shape == (39,)
    np.testing.assert_array_equal(w2, np.concatenate((w[2:-1], w[-3:-1])))
    # terminal yaw reference is fixed IN PLACE on the stored target (view semantics of the reference)
    ad.set_state([0, 0, -3.0, 5, 0, 0, 0])
    tgt = xref.copy(); tgt[:, 2] = 3.0
    mpc.set_reference(tgt, uref.copy())
    mpc.optimize()
    assert mpc.ad_opt.target[20, 2] == pytest.approx(3.0 - 2 * np.pi) and mpc.ad_opt.target[0, 2] == 3.0


def test_acados_shaped_seam_errors_and_iterate_io(tmp_path, golden_kat):
    from ad_mpc_amd.ocp_solver import AdmpcOcpSolver
    k = golden_kat
    cfg = default_config(N=k["N"], Ts=k["Ts"], terminal_scale=k["terminal_scale"])
    sol = AdmpcOcpSolver(cfg)
    with pytest.raises(Exception, match="mismatching dimension"):
        sol.set(0, "yref", np.zeros(7))
    with pytest.raises(Exception, match="not a valid argument"):
        sol.set(0, "foo", np.zeros(7))
    with pytest.raises(Exception):
        sol.get(41, "x")
    # load the reference-format iterate (same keys as sim_car_iterate.json), solve once, store, reload
    it = {}
    for i in range(k["N"] + 1):
        it["x_%d" % i] = k["X"][i]; it["u_%d" % i] = k["U"][i] if i < k["N"] else []
    f = tmp_path / "iterate.json"; f.write_text(json.dumps(it))
    sol.load_iterate(str(f))
    for j in range(k["N"]):
        sol.set(j, "yref", np.array(k["yref"][j]))
    sol.set(k["N"], "yref", np.array(k["yref_e"]))
    sol.set(0, "lbx", np.array(k["x0"])); sol.set(0, "ubx", np.array(k["x0"]))
    for j in range(k["N"] + 1):
        sol.set(j, "p", np.array([0.0]))
    assert sol.solve() == 0
    U = np.array([sol.get(i, "u") for i in range(k["N"])])
    assert np.abs(U - np.array(k["U"])).max() < 1e-8
    g = tmp_path / "out.json"
    sol.store_iterate(str(g))
    d = json.loads(g.read_text())
    assert np.allclose(d["u_0"], U[0]) and "lam_0" in d and "x_40" in d
    with pytest.raises(Exception):
        sol.store_iterate(str(g))                        # exists, overwrite=False
    sol.set(1, "p", np.array([0.5]))
    with pytest.raises(Exception, match="same on all stages"):
        sol.solve()


def test_shift_iterate_option_of_the_seam(oracle):
    """Off by default (the reference never shifts); "copy"/"rollout" shift the stored iterate before the solve."""
    from ad_mpc_amd.ocp_solver import AdmpcOcpSolver
    from ad_mpc_amd.scenarios import random_scenarios
    cfg = default_config(N=20)
    s = random_scenarios(1, N=20, seed=21)

    def run(mode):
        sol = AdmpcOcpSolver(cfg)
        assert sol.shift_iterate is None
        for k in range(20):
            sol.set(k, "yref", s["yref"][0, k])
        sol.set(20, "yref", s["yref_e"][0])
        sol.set(0, "lbx", s["x0"][0]); sol.set(0, "ubx", s["x0"][0])
        for k in range(21):
            sol.set(k, "p", np.array([s["p"][0]])); sol.set(k, "x", s["x0"][0])
        assert sol.solve() == 0
        X1 = np.array([sol.get(k, "x") for k in range(21)]); U1 = np.array([sol.get(k, "u") for k in range(20)])
        sol.shift_iterate = mode
        assert sol.solve() == 0
        return X1, U1, np.array([sol.get(k, "u") for k in range(20)])

    X1, U1, u_plain = run(None)
    for mode in ("copy", "rollout"):
        _, _, u_shift = run(mode)
        xs, us = oracle.shift_batch(cfg, X1[None], U1[None], s["p"], rollout=mode == "rollout")
        want = oracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], xs, us)[1][0]
        assert np.abs(u_shift - want).max() < 1e-8 and np.abs(u_shift - u_plain).max() > 1e-6
    sol = AdmpcOcpSolver(cfg); sol.shift_iterate = "yes"
    with pytest.raises(Exception, match="shift_iterate"):
        sol.solve()


def test_seam_adopts_the_iterate_at_the_sqp_limit(oracle):
    """acados keeps the last SQP iterate on status 2 (nlp_solver_max_iter reached); only a QP failure (status 4) leaves it
    untouched.  With sqp_iters = 2 and a tolerance the random scenario cannot reach in two steps the seam must return status 2
    AND hand back the new iterate (get / store_iterate), equal to two oracle steps."""
    from ad_mpc_amd.ocp_solver import AdmpcOcpSolver
    from ad_mpc_amd.scenarios import random_scenarios
    cfg = default_config(N=20)
    cfg.sqp_iters = 2; cfg.sqp_tol = 1e-12
    s = random_scenarios(1, N=20, seed=33)
    sol = AdmpcOcpSolver(cfg)
    for k in range(20):
        sol.set(k, "yref", s["yref"][0, k])
    sol.set(20, "yref", s["yref_e"][0])
    sol.set(0, "lbx", s["x0"][0]); sol.set(0, "ubx", s["x0"][0])
    for k in range(21):
        sol.set(k, "p", np.array([s["p"][0]])); sol.set(k, "x", s["xbar"][0, k])
    before = np.array([sol.get(k, "u") for k in range(20)])
    assert sol.solve() == 2
    after = np.array([sol.get(k, "u") for k in range(20)])
    xo, uo, co, so, io = oracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    assert so[0] == 2
    assert np.abs(after - before).max() > 1e-3                      # the iterate moved
    assert np.abs(after - uo[0]).max() < 1e-8                       # ... to the oracle's second SQP iterate
    assert sol.get_cost() == pytest.approx(co[0], rel=1e-9)        # cost and iterate belong together


def test_store_iterate_carries_the_acados_multipliers(golden_kat, tmp_path):
    """AdmpcOcpSolver.store_iterate writes pi / lam / t / sl / su in the layout of the reference's sim_car_iterate.json
    (stage 0: 22 multipliers, later stages 10, terminal stage none): solve one RTI step from the reference's converged
    acados iterate, store, and compare with the PI / LAM acados stored (<= 1e-7); a load_iterate of the file round-trips."""
    import json
    from ad_mpc_amd.ocp_solver import AdmpcOcpSolver
    k = golden_kat
    N = k["N"]
    cfg = default_config(N=N, Ts=k["Ts"], terminal_scale=k["terminal_scale"])
    sol = AdmpcOcpSolver(cfg)
    X, U = np.array(k["X"]), np.array(k["U"])
    for j in range(N + 1):
        sol.set(j, "x", X[j])
        if j < N:
            sol.set(j, "u", U[j]); sol.set(j, "yref", np.array(k["yref"][j])); sol.set(j, "p", np.array([0.0]))
    sol.set(N, "yref", np.array(k["yref_e"])); sol.set(N, "p", np.array([0.0]))
    sol.set(0, "lbx", np.array(k["x0"])); sol.set(0, "ubx", np.array(k["x0"]))
    assert sol.solve() == 0
    f = str(tmp_path / "it.json")
    sol.store_iterate(f)
    d = json.load(open(f))
    assert len(d["lam_0"]) == 22 and len(d["t_0"]) == 22 and len(d["lam_1"]) == 10 and d["lam_%d" % N] == [] and d["pi_%d" % N] == []
    for j in range(N):
        assert np.abs(np.array(d["pi_%d" % j]) - np.array(k["PI"][j])).max() <= 1e-7
        dl = np.abs(np.array(d["lam_%d" % j]) - np.array(k["LAM"][j]))
        if j == 0:
            dl[[4, 13]] = 0.0       # psi entry of the x0-equality multiplier: depends on yref_0[psi], which the fixture cannot know (see tests/test_rowqp_emu.py)
        assert dl.max() <= 1e-7
        assert np.abs(np.array(d["u_%d" % j]) - U[j]).max() <= 1e-8
        assert len(d["sl_%d" % j]) == 2 and len(d["su_%d" % j]) == 2
    sol2 = AdmpcOcpSolver(cfg)
    sol2.load_iterate(f)
    assert np.abs(sol2.get(3, "x") - sol.get(3, "x")).max() == 0.0


@pytest.mark.gpu
def test_clustered_gp_ensemble_routes_every_instance_to_its_cluster(tmp_path, oracle):
    """SURVEY 8f-4, gp.py:738-770 + quad_3d_optimizer.py:207,452: one engine handle per cluster, instances routed by the nearest
    centroid of the reference speed.  Every instance must equal the oracle solve with the GP of ITS cluster (<= 1e-8, same
    iteration counts), and must differ from the solve with another cluster's GP (the routing matters)."""
    import torch
    from test_gp_loader import _ensemble_models
    from ad_mpc_amd import gp_loader
    from ad_mpc_amd.config import set_gp
    from ad_mpc_amd.engine import EnsembleBatchSolver
    from ad_mpc_amd.scenarios import random_scenarios
    ens = gp_loader.GPEnsemble.from_pickled({"models": _ensemble_models(tmp_path)})
    cfg = default_config(N=20)
    s = random_scenarios(96, N=20, seed=17, blend=(3.0, 5.0))
    eng = EnsembleBatchSolver(cfg, ens, device=0)
    d = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    x0, yref, yref_e, p = d(s["x0"]), d(s["yref"]), d(s["yref_e"]), d(s["p"])
    xb, ub = d(s["xbar"]).clone(), d(s["ubar"]).clone()
    cost = torch.empty(96, dtype=torch.float64, device="cuda"); st = torch.empty(96, dtype=torch.int32, device="cuda"); it = torch.empty(96, dtype=torch.int32, device="cuda")
    gp_ind = eng.select(x0, ub[:, 0, :])                                   # selection on the measured state (speed feature)
    ind = gp_ind.cpu().numpy()
    np.testing.assert_array_equal(ind, ens.select_gp(ens.get_z(s["x0"], s["ubar"][:, 0, :])))
    assert len(np.unique(ind)) == 3                                          # the batch really spans the three clusters
    eng.solve(gp_ind, x0, yref, yref_e, p, xb, ub, cost, st, it)
    torch.cuda.synchronize()
    U, X = ub.cpu().numpy(), xb.cpu().numpy()
    differs = 0
    for c in range(3):
        m = ind == c
        cc = cfg.copy(); set_gp(cc, ens.clusters[c])
        o = oracle.solve_batch(cc, s["x0"][m], s["yref"][m], s["yref_e"][m], s["p"][m], s["xbar"][m], s["ubar"][m])
        np.testing.assert_array_equal(st.cpu().numpy()[m], o[3]); np.testing.assert_array_equal(it.cpu().numpy()[m], o[4])
        assert np.abs(U[m] - o[1]).max() <= 1e-8 and np.abs(X[m] - o[0]).max() <= 1e-8
        co = cfg.copy(); set_gp(co, ens.clusters[(c + 1) % 3])
        oo = oracle.solve_batch(co, s["x0"][m], s["yref"][m], s["yref_e"][m], s["p"][m], s["xbar"][m], s["ubar"][m])
        differs += int(np.abs(U[m] - oo[1]).max() > 1e-6)
    assert differs == 3
    # an instance routed to no cluster is reported as failed (status 4, cost +inf, iterate untouched), the others are solved as before
    bad = gp_ind.clone(); bad[5] = 7; bad[11] = -1
    xb2, ub2 = d(s["xbar"]).clone(), d(s["ubar"]).clone()
    eng.solve(bad, x0, yref, yref_e, p, xb2, ub2, cost, st, it)
    torch.cuda.synchronize()
    stn = st.cpu().numpy()
    assert stn[5] == 4 and stn[11] == 4 and np.isinf(cost.cpu().numpy()[[5, 11]]).all()
    keep = np.ones(96, dtype=bool); keep[[5, 11]] = False
    assert np.array_equal(ub2.cpu().numpy()[keep], U[keep]) and np.array_equal(ub2.cpu().numpy()[[5, 11]], s["ubar"][[5, 11]])
    eng.close()
