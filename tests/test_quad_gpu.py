"""Quadrotor path on the GPU (include/admpc_quad.h) against the CPU oracle, through the C ABI."""
import json
import os

import numpy as np
import pytest

from ad_mpc_amd.quad_config import default_quad_config, tight_quad_config, QNX, QNU
from ad_mpc_amd.quad_scenarios import random_quad_scenarios

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quad_shooting.json")


@pytest.fixture(scope="module")
def qoracle():
    from oracle.quad_oracle import QuadOracle
    return QuadOracle()


@pytest.fixture(scope="module")
def qeng():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ad_mpc_amd.engine import QuadBatchSolver
    return QuadBatchSolver(default_quad_config(), device=0)


def test_quad_shooting_against_the_reference_golden_vectors(qeng):
    """phi, A, B of the device against vectors from the reference's compiled CasADi code (ERK4, h = 0.1): 1e-11 relative."""
    import torch
    cases = json.load(open(GOLDEN))["cases"]
    N = qeng.cfg.N
    B = (len(cases) + N - 1) // N
    xbar = np.zeros((B, N + 1, QNX)); xbar[:, :, 3] = 1.0; ubar = np.full((B, N, QNU), 0.1)
    for i, c in enumerate(cases):
        xbar[i // N, i % N] = c["x"]; ubar[i // N, i % N] = c["u"]
    d = lambda a: torch.as_tensor(a, device="cuda")
    phi, A, Bm = qeng.shoot(d(xbar), d(ubar))
    torch.cuda.synchronize()
    phi, A, Bm = phi.cpu().numpy(), A.cpu().numpy(), Bm.cpu().numpy()
    for i, c in enumerate(cases):
        for got, ref in ((phi[i // N, i % N], c["phi"]), (A[i // N, i % N], c["A"]), (Bm[i // N, i % N], c["B"])):
            ref = np.array(ref)
            assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("N,B", [(10, 257), (10, 2048), (5, 64), (16, 40), (17, 33), (20, 300), (24, 40)])
def test_quad_solve_parity_with_oracle(qoracle, N, B):
    """One RTI step per instance: identical status and interior-point iteration counts, inputs and states within 1e-8, cost 1e-9
    relative; saturated inputs in the batch; x_0 pinned; inputs inside the box."""
    from ad_mpc_amd.engine import QuadBatchSolver
    cfg = default_quad_config(N=N, t_horizon=0.1 * N)
    eng = QuadBatchSolver(cfg, device=0)
    s = random_quad_scenarios(B, cfg, seed=100 + N)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    np.testing.assert_array_equal(g[3], o[3]); assert (o[3] == 0).all()
    np.testing.assert_array_equal(g[4], o[4])
    assert np.abs(g[1] - o[1]).max() <= 1e-8 and np.abs(g[0] - o[0]).max() <= 1e-8, (np.abs(g[1] - o[1]).max(), np.abs(g[0] - o[0]).max())
    np.testing.assert_allclose(g[2], o[2], rtol=1e-9)
    assert (g[1] >= -1e-9).all() and (g[1] <= 1 + 1e-9).all() and ((g[1] <= 1e-6) | (g[1] >= 1 - 1e-6)).sum() > B // 4
    np.testing.assert_array_equal(g[0][:, 0], s["x0"])
    eng.close()


@pytest.mark.parametrize("generic", ["0", "1"])
def test_quad_fallback_mode_on_the_device(qoracle, monkeypatch, generic):
    """ADMPC_QUAD_IPM_FALLBACK_ITER on both device paths (dense40 fast path, LDS-resident generic path): the batch that holds the known
    cycling instance (3012; 47 iterations with the fallback) -- every instance, iteration for iteration with the oracle."""
    from ad_mpc_amd.engine import QuadBatchSolver
    monkeypatch.setenv("ADMPC_QUAD_GENERIC", generic)
    cfg = tight_quad_config()
    eng = QuadBatchSolver(cfg, device=0)
    s = random_quad_scenarios(4096, cfg, seed=202)
    if generic == "1":
        s = {k: v[2900:3100] for k, v in s.items()}
    b = 3012 - (2900 if generic == "1" else 0)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    np.testing.assert_array_equal(g[3], o[3]); assert (o[3] == 0).all()
    np.testing.assert_array_equal(g[4], o[4])
    assert o[4][b] == 47
    assert np.abs(g[1] - o[1]).max() <= 1e-8 and np.abs(g[0] - o[0]).max() <= 1e-8
    eng.close()


def test_quad_failure_status_and_repeatability(qeng, qoracle):
    cfg = qeng.cfg
    s = random_quad_scenarios(48, cfg, seed=5)
    s["x0"][7, 1] = np.nan
    a = qeng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    b = qeng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)                       # bit-wise repeatable
    assert a[3][7] == 4 and np.isinf(a[2][7]) and (a[0][7] == s["xbar"][7]).all() and (a[1][7] == s["ubar"][7]).all()
    o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    np.testing.assert_array_equal(a[3], o[3])


def test_quad_3d_optimizer_constructs_with_the_reference_defaults(qoracle):
    """The reference class defaults to n_nodes = 20, t_horizon = 1 (quad_3d_optimizer.py:29): 80 inputs, more than one per lane of a
    wavefront.  Round 2 refused it at admpc_quad_create; it now runs on the two-wave kernel (horizons 17..24) and follows the oracle."""
    from ad_mpc_amd.quad_3d_optimizer import Quad3DOptimizer
    opt = Quad3DOptimizer()
    assert opt.N == 20 and opt.cfg.N == 20
    x0 = np.array([0.3, -0.2, 1.0, 1, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.1])
    opt.set_reference_state([[1.0, 0.5, 1.5], [1, 0, 0, 0], [0, 0, 0], [0, 0, 0]])
    w, x = opt.run_optimization(x0, return_x=True)
    assert w.shape == (80,) and x.shape == (21, 13) and np.isfinite(w).all() and w.min() >= -1e-9 and w.max() <= 1 + 1e-9
    np.testing.assert_allclose(x[0], x0, atol=1e-12)


def test_quad_argument_errors():
    import ctypes as C
    from ad_mpc_amd import _lib
    L = _lib.load()
    bad = default_quad_config(); bad.N = 40
    h = C.c_void_p(0)
    assert L.admpc_quad_create(C.byref(bad), 0, C.byref(h)) == -1 and b"N must be" in L.admpc_last_error()
    bad = default_quad_config(); bad.W[14] = 0.0
    assert L.admpc_quad_create(C.byref(bad), 0, C.byref(h)) == -1
    assert L.admpc_quad_create(C.byref(default_quad_config()), 99, C.byref(h)) == -2


def test_quad_3d_optimizer_shim_against_oracle(qoracle):
    """The reference-shaped host class: weight expansion with q_mask (my_quad_acados_ocp.json has W[3] = 0), body-frame velocity in the
    point reference, padding of a short trajectory, persistent iterate over two calls -- each call equals an oracle step from the
    same data."""
    from ad_mpc_amd.quad_3d_optimizer import Quad3DOptimizer, v_dot_q, quaternion_inverse
    opt = Quad3DOptimizer(None, t_horizon=1.0, n_nodes=10, q_mask=np.ones(12))
    np.testing.assert_allclose(np.array(opt.cfg.W[:]), [10, 10, 10, 0, .1, .1, .1] + [.05] * 6 + [.1] * 4)
    assert bytes(opt.cfg) == bytes(default_quad_config())                        # exactly the shipped OCP
    q = np.array([0.9, 0.1, -0.2, 0.3]); q /= np.linalg.norm(q)
    tgt = [[1.0, -2.0, 0.5], list(q), [0.5, 0.2, -0.1], [0, 0, 0]]
    opt.set_reference_state(tgt, [0.12] * 4)
    np.testing.assert_allclose(opt.yref[0][3, 7:10], v_dot_q(tgt[2], quaternion_inverse(q)))      # the reference's body-frame quirk
    x0 = [0.2, 0.1, -0.3, 1, 0, 0, 0, 0.1, 0, 0, 0, 0, 0.1]
    xi, ui = opt.x_iter[0].copy(), opt.u_iter[0].copy()
    for _ in range(2):
        w, x = opt.run_optimization(x0, return_x=True)
        o = qoracle.solve_batch(opt.cfg, np.array(x0)[None], opt.yref[0][None], opt.yref_e[0][None], xi[None], ui[None])
        assert o[3][0] == 0 and np.abs(w - o[1][0].reshape(-1)).max() <= 1e-8 and np.abs(x - o[0][0]).max() <= 1e-8
        xi, ui = o[0][0], o[1][0]
    # trajectory reference shorter than the horizon: padded with its last row; the terminal node takes the state only
    T = 6
    xt = [np.linspace(0, 1, T)[:, None] * np.ones((1, 3)), np.tile([1.0, 0, 0, 0], (T, 1)), np.zeros((T, 3)), np.zeros((T, 3))]
    opt.set_reference_trajectory(xt, np.full((T - 1, 4), 0.12))
    assert (opt.yref[0][T:, :3] == 1.0).all() and (opt.yref_e[0][:3] == 1.0).all() and opt.yref[0].shape == (10, 17)
    w = opt.run_optimization(x0)
    assert w.shape == (40,) and opt.status == 0


def test_quad_gp_residual_on_the_device(qoracle):
    """GP residual of the body-frame acceleration (quad_3d_optimizer.py:289-327): device shooting against the oracle at 1e-11, the
    solve with identical iteration counts at 1e-8, and the GP really changes the answer."""
    import torch
    from ad_mpc_amd.engine import QuadBatchSolver
    from ad_mpc_amd.quad_config import set_quad_gp
    from test_quad_oracle import quad_gps
    cfg = default_quad_config(); set_quad_gp(cfg, quad_gps())
    eng = QuadBatchSolver(cfg, device=0)
    s = random_quad_scenarios(96, cfg, seed=31)
    d = lambda a: torch.as_tensor(a, device="cuda")
    phi, A, Bm = eng.shoot(d(s["xbar"]), d(s["ubar"]))
    torch.cuda.synchronize()
    phi, A, Bm = phi.cpu().numpy(), A.cpu().numpy(), Bm.cpu().numpy()
    for b in range(0, 96, 11):
        for k in (0, 4, 9):
            po, Ao, Bo = qoracle.rk4_sens(cfg, s["xbar"][b, k], s["ubar"][b, k], cfg.Ts, gpx=s["xbar"][b, 0] if k == 0 else None)   # node 0: GP-state parameter
            for got, ref in ((phi[b, k], po), (A[b, k], Ao), (Bm[b, k], Bo)):
                assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    np.testing.assert_array_equal(g[3], o[3]); np.testing.assert_array_equal(g[4], o[4]); assert (o[3] == 0).all()
    assert np.abs(g[1] - o[1]).max() <= 1e-8 and np.abs(g[0] - o[0]).max() <= 1e-8
    base = qoracle.solve_batch(default_quad_config(), s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    assert np.abs(base[1] - o[1]).max() > 1e-3
    eng.close()


def test_first_node_gp_state_on_the_device(qoracle):
    """quad_3d_optimizer.py:291-297, :546-552 on the device: the first node evaluates the GP residual at a per-instance PARAMETER
    (admpc_quad_solve_batch_ex's gp_state; default: the initial state).  Shooting of node 0 with a foreign GP state against the oracle
    (the state columns of A_0 carry no GP term), solves with gp_state = None / = x0 identical and equal to the oracle's, a different
    gp_state follows the oracle's solve with that state -- identical iteration counts, 1e-8."""
    import torch
    from ad_mpc_amd.engine import QuadBatchSolver
    from ad_mpc_amd.quad_config import set_quad_gp
    from test_quad_oracle import quad_gps
    cfg = default_quad_config(); set_quad_gp(cfg, quad_gps())
    eng = QuadBatchSolver(cfg, device=0)
    B = 64
    s = random_quad_scenarios(B, cfg, seed=41)
    rng = np.random.default_rng(2)
    gs = s["x0"] + 0.5 * rng.standard_normal((B, QNX)); gs[:, 3:7] /= np.linalg.norm(gs[:, 3:7], axis=1, keepdims=True)
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    phi, A, Bm = eng.shoot(d(s["xbar"]), d(s["ubar"]), gp_state=d(gs))
    torch.cuda.synchronize()
    phi, A, Bm = phi.cpu().numpy(), A.cpu().numpy(), Bm.cpu().numpy()
    nominal = default_quad_config()
    for b in range(0, B, 9):
        po, Ao, Bo = qoracle.rk4_sens(cfg, s["xbar"][b, 0], s["ubar"][b, 0], cfg.Ts, gpx=gs[b])
        for got, ref in ((phi[b, 0], po), (A[b, 0], Ao), (Bm[b, 0], Bo)):
            assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        assert np.abs(A[b, 0] - qoracle.rk4_sens(nominal, s["xbar"][b, 0], s["ubar"][b, 0], cfg.Ts)[1]).max() <= 1e-11
        po, Ao, Bo = qoracle.rk4_sens(cfg, s["xbar"][b, 3], s["ubar"][b, 3], cfg.Ts)                  # later nodes: the integrated state
        assert np.abs(A[b, 3] - Ao).max() <= 1e-11 * max(1.0, np.abs(Ao).max())
    g0 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    g1 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], gp_state=s["x0"])
    g2 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], gp_state=gs)
    for a, b_ in zip(g0, g1):
        np.testing.assert_array_equal(a, b_)
    for g, kw in ((g0, {}), (g2, dict(gp_state=gs))):
        o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8, **kw)
        np.testing.assert_array_equal(g[3], o[3]); np.testing.assert_array_equal(g[4], o[4]); assert (o[3] == 0).all()
        assert np.abs(g[1] - o[1]).max() <= 1e-8 and np.abs(g[0] - o[0]).max() <= 1e-8
    assert np.abs(g0[1] - g2[1]).max() > 1e-6
    eng.close()


@pytest.mark.parametrize("N", [10, 20])
def test_linear_drag_on_the_device(qoracle, N):
    """cfg.rdrv (quad_3d_optimizer.py:364-381): device shooting and solves with the drag term against the oracle, on the one-wave kernel
    and on the two-wave kernel of the class's default horizon; the term really changes the answer."""
    import torch
    from ad_mpc_amd.engine import QuadBatchSolver
    cfg = default_quad_config(N=N); cfg.rdrv[0], cfg.rdrv[1], cfg.rdrv[2] = -0.35, -0.25, -0.1
    eng = QuadBatchSolver(cfg, device=0)
    B = 48
    s = random_quad_scenarios(B, cfg, seed=51)
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    phi, A, Bm = eng.shoot(d(s["xbar"]), d(s["ubar"]))
    torch.cuda.synchronize()
    phi, A, Bm = phi.cpu().numpy(), A.cpu().numpy(), Bm.cpu().numpy()
    for b in range(0, B, 7):
        for k in (0, N // 2, N - 1):
            po, Ao, Bo = qoracle.rk4_sens(cfg, s["xbar"][b, k], s["ubar"][b, k], cfg.Ts)
            for got, ref in ((phi[b, k], po), (A[b, k], Ao), (Bm[b, k], Bo)):
                assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    np.testing.assert_array_equal(g[3], o[3]); np.testing.assert_array_equal(g[4], o[4]); assert (o[3] == 0).all()
    assert np.abs(g[1] - o[1]).max() <= 1e-8 and np.abs(g[0] - o[0]).max() <= 1e-8
    base = qoracle.solve_batch(default_quad_config(N=N), s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    assert np.abs(base[1] - o[1]).max() > 1e-4
    eng.close()


def test_cluster_routing_of_the_quadrotor_ensemble(qoracle):
    """One solver per GP cluster, chosen per solve from the reference state (quad_3d_optimizer.py:207, :446-452, :485-491; gp.py:738-770):
    admpc_quad_select_cluster_batch against numpy's nearest centroid on the body-frame features (ties to the lower index), and
    admpc_quad_solve_batch_routed: every instance equals the oracle's solve with the GPs of ITS cluster and differs from a neighbour's;
    an out-of-range route fails with the iterate untouched.  The host class keeps one iterate per cluster as the reference's solvers do."""
    import torch
    from ad_mpc_amd.engine import QuadEnsembleBatchSolver
    from ad_mpc_amd.quad_config import set_quad_gp
    from ad_mpc_amd.quad_3d_optimizer import Quad3DOptimizer, QuadGPEnsemble, q_to_rot_mat
    from test_quad_oracle import quad_gps
    clusters = [quad_gps(seed=1), quad_gps(seed=2), quad_gps(seed=3)]
    cent = np.array([[-1.0, 0.25], [0.5, 0.5], [2.0, 0.75]]); feats = [7, 13]          # body-frame v_x and the first rotor's input
    cfg = default_quad_config()
    ens = QuadEnsembleBatchSolver(cfg, clusters, cent, feats, device=0)
    B = 96
    s = random_quad_scenarios(B, cfg, seed=61)
    rng = np.random.default_rng(6)
    xs = s["x0"] + rng.standard_normal((B, QNX)); xs[:, 3:7] /= np.linalg.norm(xs[:, 3:7], axis=1, keepdims=True)
    us = rng.uniform(0, 1, (B, QNU))
    xs[0, 3:7] = [1, 0, 0, 0]; xs[0, 7] = -0.25; us[0, 0] = 0.375                    # equidistant from centroids 0 and 1: the lower index
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    route = ens.select(d(xs), d(us))
    torch.cuda.synchronize()
    route = route.cpu().numpy()
    vb = np.array([q_to_rot_mat(xs[b, 3:7]).T @ xs[b, 7:10] for b in range(B)])
    z = np.c_[vb[:, 0], us[:, 0]]
    want = np.argmin(np.sqrt(((z[:, None, :] - cent[None]) ** 2).sum(2)), axis=1)
    np.testing.assert_array_equal(route, want)
    assert route[0] == 0 and len(np.unique(route)) == 3
    r2 = route.copy(); r2[5] = 7; r2[6] = -1
    xb, ub = d(s["xbar"]).clone(), d(s["ubar"]).clone()
    cost = torch.empty(B, dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
    ens.solve(d(r2.astype(np.int32)), d(s["x0"]), d(s["yref"]), d(s["yref_e"]), xb, ub, cost, st, it)
    torch.cuda.synchronize()
    xg, ug, stg, itg, cg = xb.cpu().numpy(), ub.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), cost.cpu().numpy()
    assert stg[5] == 4 and stg[6] == 4 and np.isinf(cg[[5, 6]]).all() and np.array_equal(ug[[5, 6]], s["ubar"][[5, 6]]) and np.array_equal(xg[[5, 6]], s["xbar"][[5, 6]])
    sol = []
    for c in range(3):
        cc = cfg.copy(); set_quad_gp(cc, clusters[c])
        sol.append(qoracle.solve_batch(cc, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8))
    ok = np.ones(B, dtype=bool); ok[[5, 6]] = False
    for b in np.nonzero(ok)[0]:
        o = sol[route[b]]
        assert stg[b] == o[3][b] == 0 and itg[b] == o[4][b] and np.abs(ug[b] - o[1][b]).max() <= 1e-8 and np.abs(xg[b] - o[0][b]).max() <= 1e-8
    other = np.array([np.abs(ug[b] - sol[(route[b] + 1) % 3][1][b]).max() for b in np.nonzero(ok)[0]])
    assert np.median(other) > 1e-5
    ens.close()
    # the host class: set_reference_state returns the cluster of the target, run_optimization(use_model=...) runs that cluster's solver
    opt = Quad3DOptimizer(None, t_horizon=1.0, n_nodes=10, gp_regressors=QuadGPEnsemble(clusters, cent, feats))
    tgt = [[1.0, -2.0, 0.5], [1.0, 0, 0, 0], [2.2, 0.0, 0.0], [0, 0, 0]]
    ind = opt.set_reference_state(tgt, [0.9, 0.2, 0.2, 0.2])
    assert ind == 2
    x0 = [0.2, 0.1, -0.3, 1, 0, 0, 0, 0.1, 0, 0, 0, 0, 0.1]
    gst = [0.0, 0, 0, 1, 0, 0, 0, 0.6, -0.2, 0.1, 0, 0, 0]
    w, x = opt.run_optimization(x0, use_model=ind, return_x=True, gp_regression_state=gst)
    cc = cfg.copy(); set_quad_gp(cc, clusters[2])
    o = qoracle.solve_batch(cc, np.array(x0)[None], opt.yref[2][None], opt.yref_e[2][None], np.zeros((1, 11, QNX)), np.zeros((1, 10, QNU)), gp_state=np.array(gst)[None])
    assert opt.status == 0 and np.abs(w - o[1][0].reshape(-1)).max() <= 1e-8 and np.abs(x - o[0][0]).max() <= 1e-8
    assert (opt.x_iter[0] == 0).all() and (opt.x_iter[2] != 0).any()                # the other clusters' iterates did not move


def test_quad_sqp_mode_on_the_device(qoracle):
    """solver_type "SQP" (create_ros_gp_mpc.py:63-68 -> quad_3d_optimizer.py:203): cfg.sqp_iters QPs per call, acados' stopping test on the four
    KKT residual norms in front of every QP but the first (admpc_quad_sqp_test_kernel), status 2 at the limit.  Device against the oracle:
    identical statuses, iterates within 1e-7 (the converged ones within 1e-6 of each other's fixed point: up to a hundred QPs of rounding
    apart); plain multi-step mode (tolerance 0) bit-for-bit equal to that many single calls; the host class maps and refuses solver types."""
    from ad_mpc_amd.engine import QuadBatchSolver
    from ad_mpc_amd.quad_3d_optimizer import Quad3DOptimizer
    cfg = default_quad_config()
    s = random_quad_scenarios(96, cfg, seed=7, pos_err=0.8, tilt=0.2, aggressive=0.0)
    for iters, tol in ((100, 1e-6), (4, 1e-6), (3, 0.0)):
        c = cfg.copy(); c.sqp_iters, c.sqp_tol = iters, tol
        eng = QuadBatchSolver(c, device=0)
        g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
        o = qoracle.solve_batch(c, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
        np.testing.assert_array_equal(g[3], o[3])
        assert set(o[3].tolist()) <= {0, 2}
        if tol > 0 and iters == 100:
            assert (o[3] == 0).sum() >= 30 and (o[3] == 2).sum() >= 10
        lim = 1e-6 if iters == 100 else 1e-8
        assert np.abs(g[1] - o[1]).max() <= lim and np.abs(g[0] - o[0]).max() <= 10 * lim, (iters, np.abs(g[1] - o[1]).max(), np.abs(g[0] - o[0]).max())
        eng.close()
    # three RTI steps in one call = three calls
    c3 = cfg.copy(); c3.sqp_iters, c3.sqp_tol = 3, 0.0
    e1 = QuadBatchSolver(cfg, device=0); e3 = QuadBatchSolver(c3, device=0)
    xa, ua = s["xbar"], s["ubar"]
    for _ in range(3):
        xa, ua, *_ = e1.solve_numpy(s["x0"], s["yref"], s["yref_e"], xa, ua)
    xb, ub, _, stb, _ = e3.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    np.testing.assert_array_equal(xa, xb); np.testing.assert_array_equal(ua, ub); assert (stb == 0).all()
    e1.close(); e3.close()
    # the host class: "SQP" is honoured, anything unknown is refused
    opt = Quad3DOptimizer(solver_options={"solver_type": "SQP", "terminal_cost": True})
    assert opt.cfg.sqp_iters == 100 and opt.cfg.sqp_tol == 1e-6 and opt.cfg.We[0] > 0
    opt.set_reference_state(x_target=[[0.3, -0.2, 0.4], [1, 0, 0, 0], [0, 0, 0], [0, 0, 0]])
    w = opt.run_optimization(initial_state=[0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    assert opt.status in (0, 2) and np.isfinite(w).all()
    rti = Quad3DOptimizer(solver_options={"solver_type": "SQP_RTI", "terminal_cost": False})
    assert rti.cfg.sqp_iters == 1
    with pytest.raises(ValueError):
        Quad3DOptimizer(solver_options={"solver_type": "DDP", "terminal_cost": False})


def test_quad_segmented_kernel_at_the_class_default_horizon(qoracle, monkeypatch):
    """N = 20 (quad_3d_optimizer.py:28) runs on admpc_quad_seg_kernel: two cooperating waves per instance.  Beyond the parity cases above: a batch
    large enough for the work counter (B > 8 x the grid), a non-finite instance (status 4, iterate untouched, its neighbours unaffected), bit-wise
    repeatability, a routed mask, and agreement with the dense two-wave kernel of round 3 (ADMPC_QUAD_WIDE=1) on the same inputs."""
    from ad_mpc_amd.engine import QuadBatchSolver
    cfg = default_quad_config(N=20, t_horizon=2.0)
    B = 5000
    s = random_quad_scenarios(B, cfg, seed=77)
    s["x0"][123, 8] = np.nan
    eng = QuadBatchSolver(cfg, device=0)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=16)
    np.testing.assert_array_equal(g[3], o[3]); assert g[3][123] == 4 and (np.delete(g[3], 123) == 0).all() and np.isinf(g[2][123])
    np.testing.assert_array_equal(g[0][123], s["xbar"][123]); np.testing.assert_array_equal(g[1][123], s["ubar"][123])
    ok = o[3] == 0
    np.testing.assert_array_equal(g[4][ok], o[4][ok])
    assert np.abs(g[1][ok] - o[1][ok]).max() <= 1e-8 and np.abs(g[0][ok] - o[0][ok]).max() <= 1e-8
    np.testing.assert_allclose(g[2][ok], o[2][ok], rtol=1e-9)
    g2 = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    for a, b in zip(g, g2):
        np.testing.assert_array_equal(a, b)
    eng.close()
    monkeypatch.setenv("ADMPC_QUAD_WIDE", "1")
    sub = {k: v[:96] for k, v in s.items()}
    ew = QuadBatchSolver(cfg, device=0)
    gw = ew.solve_numpy(sub["x0"], sub["yref"], sub["yref_e"], sub["xbar"], sub["ubar"])
    ew.close()
    np.testing.assert_array_equal(gw[4], g[4][:96])
    assert np.abs(gw[1] - g[1][:96]).max() <= 1e-8 and (gw[1] != g[1][:96]).any()          # another kernel, another elimination order: close, not bit-equal
