"""Quadrotor path on the GPU (include/admpc_quad.h) against the CPU oracle, through the C ABI."""
import json
import os

import numpy as np
import pytest

from ad_mpc_amd.quad_config import default_quad_config, QNX, QNU
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
    cfg = default_quad_config()
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
    np.testing.assert_allclose(opt.yref[3, 7:10], v_dot_q(tgt[2], quaternion_inverse(q)))      # the reference's body-frame quirk
    x0 = [0.2, 0.1, -0.3, 1, 0, 0, 0, 0.1, 0, 0, 0, 0, 0.1]
    xi, ui = opt.x_iter.copy(), opt.u_iter.copy()
    for _ in range(2):
        w, x = opt.run_optimization(x0, return_x=True)
        o = qoracle.solve_batch(opt.cfg, np.array(x0)[None], opt.yref[None], opt.yref_e[None], xi[None], ui[None])
        assert o[3][0] == 0 and np.abs(w - o[1][0].reshape(-1)).max() <= 1e-8 and np.abs(x - o[0][0]).max() <= 1e-8
        xi, ui = o[0][0], o[1][0]
    # trajectory reference shorter than the horizon: padded with its last row; the terminal node takes the state only
    T = 6
    xt = [np.linspace(0, 1, T)[:, None] * np.ones((1, 3)), np.tile([1.0, 0, 0, 0], (T, 1)), np.zeros((T, 3)), np.zeros((T, 3))]
    opt.set_reference_trajectory(xt, np.full((T - 1, 4), 0.12))
    assert (opt.yref[T:, :3] == 1.0).all() and (opt.yref_e[:3] == 1.0).all() and opt.yref.shape == (10, 17)
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
            po, Ao, Bo = qoracle.rk4_sens(cfg, s["xbar"][b, k], s["ubar"][b, k], cfg.Ts)
            for got, ref in ((phi[b, k], po), (A[b, k], Ao), (Bm[b, k], Bo)):
                assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"])
    o = qoracle.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    np.testing.assert_array_equal(g[3], o[3]); np.testing.assert_array_equal(g[4], o[4]); assert (o[3] == 0).all()
    assert np.abs(g[1] - o[1]).max() <= 1e-8 and np.abs(g[0] - o[0]).max() <= 1e-8
    base = qoracle.solve_batch(default_quad_config(), s["x0"], s["yref"], s["yref_e"], s["xbar"], s["ubar"], nthreads=8)
    assert np.abs(base[1] - o[1]).max() > 1e-3
    eng.close()
