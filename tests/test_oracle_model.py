"""Oracle pin 1: model, ERK4 and sensitivities against golden vectors produced by the REFERENCE's own
compiled CasADi code (tests/golden/shooting.json, generator oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from ad_mpc_amd.config import default_config


def test_model_and_rk4_against_reference_golden(oracle, golden_shooting):
    cfg = default_config()
    cases = golden_shooting["cases"]
    assert len(cases) == 120
    for c in cases:
        x, u, p, h = np.array(c["x"]), np.array(c["u"]), c["p"], c["h"]
        f = oracle.f(cfg, x, u, p)
        np.testing.assert_allclose(f, np.array(c["xdot"]), rtol=1e-13, atol=1e-13)
        phi, A, B = oracle.rk4_sens(cfg, x, u, p, h)
        np.testing.assert_allclose(phi, np.array(c["phi"]), rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(A, np.array(c["A"]), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(B, np.array(c["B"]), rtol=1e-11, atol=1e-13)


def test_jacobian_matches_finite_differences(oracle):
    cfg = default_config()
    rng = np.random.default_rng(5)
    for i in range(30):
        p = [0.0, 0.4, 1.0][i % 3]
        x = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-3, 3), rng.uniform(2, 15), rng.uniform(-0.5, 0.5),
                      rng.uniform(-0.5, 0.5), rng.uniform(-0.4, 0.4)])
        u = np.array([rng.uniform(-5, 5), rng.uniform(-2, 2)])
        Jx, Ju = oracle.jac(cfg, x, u, p)
        for j in range(7):
            e = np.zeros(7); e[j] = 1e-6
            fd = (oracle.f(cfg, x + e, u, p) - oracle.f(cfg, x - e, u, p)) / 2e-6
            np.testing.assert_allclose(Jx[:, j], fd, rtol=1e-6, atol=1e-5)
        for j in range(2):
            e = np.zeros(2); e[j] = 1e-6
            fd = (oracle.f(cfg, x, u + e, p) - oracle.f(cfg, x, u - e, p)) / 2e-6
            np.testing.assert_allclose(Ju[:, j], fd, rtol=1e-6, atol=1e-5)


def test_structural_facts_of_the_model(oracle):
    """Columns 0,1 of A are unit vectors, row 6 of [A B] is [e6, 0, h] (delta' = u1): the kernel relies on both."""
    cfg = default_config()
    rng = np.random.default_rng(11)
    for p in (0.0, 1.0):
        x = np.array([1.0, -2.0, 0.7, 8.0, 0.2, -0.1, 0.15]); u = rng.uniform(-1, 1, 2)
        phi, A, B = oracle.rk4_sens(cfg, x, u, p, 0.05)
        np.testing.assert_array_equal(A[:, 0], np.eye(7)[:, 0]); np.testing.assert_array_equal(A[:, 1], np.eye(7)[:, 1])
        np.testing.assert_allclose(A[6], np.eye(7)[6], atol=1e-16)
        np.testing.assert_allclose(B[6], [0.0, 0.05], atol=1e-16)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libsimcar_ref.so")),
                    reason="oracle/_ref not built (only possible where /root/reference exists)")
def test_oracle_against_live_reference_model(oracle):
    """Stronger than the committed vectors: 300 fresh random points against the compiled reference code."""
    from oracle.oracle import RefModel
    ref = RefModel(); cfg = default_config()
    rng = np.random.default_rng(123)
    for i in range(300):
        p = [0.0, 0.37, 1.0][i % 3]
        x = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-np.pi, np.pi), rng.uniform(0.5, 20), rng.uniform(-1, 1),
                      rng.uniform(-1, 1), rng.uniform(-0.5, 0.5)])
        u = np.array([rng.uniform(-10, 5), rng.uniform(-3, 3)])
        np.testing.assert_allclose(oracle.f(cfg, x, u, p), ref.ode(x, u, p), rtol=1e-13, atol=1e-13)
        phi, A, B = oracle.rk4_sens(cfg, x, u, p, 0.05); phi2, A2, B2 = ref.rk4_sens(x, u, p, 0.05)
        np.testing.assert_allclose(phi, phi2, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(A, A2, rtol=1e-11, atol=1e-13); np.testing.assert_allclose(B, B2, rtol=1e-11, atol=1e-13)
