"""local reference generator (RefTrajectory.get_waypoints).  The numpy restatement is pinned by vectors the
reference module itself produced (tests/golden/ref_traj.json); the GPU kernel is checked against both."""
import json
import os

import numpy as np
import pytest

from oracle.ref_traj_oracle import get_waypoints

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_traj.json")
KEYS = ("x_ref", "y_ref", "psi_ref", "v_ref", "cdist_ref", "curv_ref")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def test_oracle_reproduces_reference_vectors(golden):
    n_stop = 0
    for c in golden["cases"]:
        t = np.array(c["table"])
        for p in c["poses"]:
            o = get_waypoints(t, c["H"], c["dt"], p["X"], p["Y"], p["psi"])
            for k, v in p["out"].items():
                if k == "stop":
                    assert bool(v) == o[k]; n_stop += bool(v)
                else:
                    np.testing.assert_allclose(o[k], np.array(v), rtol=0, atol=1e-13)
            assert len(o["x_ref"]) == c["H"] and len(o["v_ref"]) == c["H"]
    assert n_stop > 0          # the end-of-path branch is exercised


@pytest.mark.gpu
def test_gpu_waypoints_against_reference_vectors_and_oracle(golden):
    import torch
    from ad_mpc_amd.ref_traj import RefTrajectory
    rng = np.random.default_rng(3)
    for c in golden["cases"]:
        rt = RefTrajectory(traj_horizon=c["H"], traj_dt=c["dt"])
        rt.set_traj(np.array(c["x"]), np.array(c["y"]), np.array(c["psi"]), np.array(c["vel"]))
        np.testing.assert_allclose(rt.trajectory, np.array(c["table"]), rtol=0, atol=1e-12)      # set_traj incl. curvature filter
        for p in c["poses"]:                                                                    # the reference's single-pose call
            w = rt.get_waypoints(p["X"], p["Y"], p["psi"])
            for k, v in p["out"].items():
                if k == "stop":
                    assert bool(v) == w[k]
                else:
                    np.testing.assert_allclose(w[k], np.array(v), rtol=0, atol=1e-12)
        # batched poses against the numpy restatement
        B = 257
        t = rt.trajectory
        idx = rng.integers(0, t.shape[0], B)
        X = t[idx, 1] + rng.normal(0, 1.0, B); Y = t[idx, 2] + rng.normal(0, 1.0, B); P = rng.uniform(-10, 10, B)
        dev = torch.device("cuda", 0)
        ref, err, stop = rt.get_waypoints_batch(*(torch.as_tensor(a, dtype=torch.float64, device=dev) for a in (X, Y, P)))
        torch.cuda.synchronize()
        ref, err, stop = ref.cpu().numpy(), err.cpu().numpy(), stop.cpu().numpy()
        for b in range(B):
            o = get_waypoints(t, c["H"], c["dt"], X[b], Y[b], P[b])
            for i, k in enumerate(KEYS):
                np.testing.assert_allclose(ref[b, i], o[k], rtol=0, atol=1e-12)
            np.testing.assert_allclose(err[b], [o["s0"], o["e_y0"], o["e_psi0"]], rtol=0, atol=1e-12)
            assert bool(stop[b]) == o["stop"]


@pytest.mark.gpu
def test_gpu_waypoints_argument_errors():
    import ctypes as C
    from ad_mpc_amd import _lib
    L = _lib.load()
    z = C.c_void_p(0)
    assert L.admpc_waypoints_batch(0, 100, 2, 0.05, 4, *([z] * 13), z) == -1        # H < 3
    assert L.admpc_waypoints_batch(0, 100, 65, 0.05, 4, *([z] * 13), z) == -1       # H > 64
    assert L.admpc_waypoints_batch(0, 100, 20, 0.05, 4, *([z] * 13), z) == -1       # null arrays
    assert L.admpc_waypoints_batch(0, 100, 20, 0.05, 0, *([z] * 13), z) == 0        # empty batch
