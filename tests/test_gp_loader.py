"""GP on-disk format: round trip through the reference's joblib dict layout, directory scheme, rejections, and
the loaded model against a direct numpy evaluation of the reference's predictive mean (gp.py:81-138, 446-471)."""
import os

import numpy as np
import pytest

from ad_mpc_amd.config import default_config, NX
from ad_mpc_amd import gp_loader
from ad_mpc_amd.scenarios import grid_gp


def _ref_mean(d, z):
    """mu(z) = K(z, x_train) k_inv_y + y_mean with the reference's kernel  sigma_f exp(-0.5 |z-x|^2 / l^2)."""
    x = np.asarray(d["x_train"]).reshape(-1)
    ell = float(np.squeeze(d["kernel_params"]["l"])); sf = float(d["kernel_params"]["sigma_f"])
    k = sf * np.exp(-0.5 * (z - x) ** 2 / ell ** 2)
    return float(k @ np.asarray(d["k_inv_y"]).reshape(-1) + float(np.squeeze(d["y_mean"])))


def test_round_trip_and_directory_scheme(tmp_path, oracle):
    opts = {"git": "abc123", "model_name": "car_gp", "params": {"payload": False, "drag": True}}
    directory, fname = gp_loader.get_model_dir_and_file(opts, str(tmp_path))
    assert directory == os.path.join(str(tmp_path), "abc123", "car_gp") and fname == "drag__no_payload"
    entries = grid_gp()
    saved = [gp_loader.save_regressor(os.path.join(directory, "%s_%d.pkl" % (fname, i)), e) for i, e in enumerate(entries)]
    assert set(saved[0].keys()) == set(gp_loader.SAVED_KEYS)
    open(os.path.join(directory, "feats.csv"), "w").write("ignored\n")
    pm = gp_loader.load_pickled_models(model_options=opts, save_dir=str(tmp_path))
    assert pm is not None and len(pm["models"]) == len(entries)
    cfg = default_config(N=20)
    assert gp_loader.install_from_directory(cfg, model_options=opts, save_dir=str(tmp_path)) == len(entries)
    assert cfg.n_gp == len(entries)
    # the installed model reproduces the reference's predictive mean inside the oracle's dynamics
    base = default_config(N=20)
    x = np.array([1.0, -2.0, 0.3, 6.0, 0.2, -0.1, 0.05]); u = np.array([0.5, -0.2])
    f0 = oracle.f(base, x, u, 1.0); f1 = oracle.f(cfg, x, u, 1.0)
    for d in pm["models"]:
        feat = (d["x_features"] + [NX + i for i in d["u_features"]])[0]
        z = np.concatenate([x, u])[feat]
        assert abs((f1[d["reg_dim"]] - f0[d["reg_dim"]]) - _ref_mean(d, z)) <= 1e-12
    assert gp_loader.load_pickled_models(directory=os.path.join(str(tmp_path), "nope")) is None
    assert gp_loader.load_pickled_models(directory=directory, file_name="other") is None


def test_rejections(tmp_path):
    e = grid_gp()[0]
    d = gp_loader.save_regressor(str(tmp_path / "a.pkl"), e)
    two = dict(d); two["x_features"] = [3, 4]                  # two features but one-column training inputs
    with pytest.raises(ValueError):
        gp_loader.gp_entry_from_saved(two)
    four = dict(d); four["x_features"] = [3, 4, 5, 6]; four["x_train"] = np.zeros((5, 4)); four["k_inv_y"] = np.zeros((5, 1))
    with pytest.raises(ValueError):                             # more features than the device holds
        gp_loader.gp_entry_from_saved(four)
    pos = dict(d); pos["x_features"] = [0]                      # position is not a feature of the device model
    with pytest.raises(ValueError):
        gp_loader.gp_entry_from_saved(pos)
    wrong = dict(d); wrong["kernel_type"] = "matern"
    with pytest.raises(ValueError):
        gp_loader.gp_entry_from_saved(wrong)
    big = dict(d); big["x_train"] = np.zeros((40, 1)); big["k_inv_y"] = np.zeros((40, 1))
    with pytest.raises(ValueError):
        gp_loader.gp_entry_from_saved(big)
    with pytest.raises(ValueError):                      # two regressors for the same output = clustered ensemble
        gp_loader.gps_from_pickled([d, d])
    with pytest.raises(ValueError):
        gp_loader.gp_entry_from_saved({"x_train": 1})


def _ensemble_models(tmp_path, centroids=(6.0, 2.0, 10.0), dims=(3, 4)):
    """Three clusters per output dimension, saved in a shuffled order, each cluster with its own training data."""
    rng = np.random.default_rng(5)
    saved = []
    for d in dims:
        for j, c in enumerate(centroids):
            Z = c + np.linspace(-2.0, 2.0, 9)
            e = dict(feat=3, out=d, Z=Z, alpha=0.05 * rng.standard_normal(9), length_scale=1.5, sigma_f=0.8, ymean=0.01 * j, centroid=[c])
            saved.append(gp_loader.save_regressor(str(tmp_path / ("m_%d_%d.pkl" % (d, j))), e))
    return saved


def test_ensemble_grouping_and_nearest_centroid_selection(tmp_path):
    """gp.py:575-607 (clusters sorted by their centroid, same centroids in every dimension) and gp.py:738-770 (select_gp) on a
    hand-derived fixture: centroids 2, 6, 10 -> boundaries at 4 and 8, a tie goes to the lower index (numpy.argmin)."""
    saved = _ensemble_models(tmp_path)
    ens = gp_loader.GPEnsemble.from_pickled({"models": saved})
    assert ens.n_models == 3 and ens.feat == 3
    np.testing.assert_array_equal(ens.centroids[:, 0], [2.0, 6.0, 10.0])
    for c, cen in enumerate((2.0, 6.0, 10.0)):                 # cluster c carries the regressors trained around centroid c, one per dimension
        assert [g["out"] for g in ens.clusters[c]] == [3, 4]
        for g in ens.clusters[c]:
            assert abs(np.mean(g["Z"]) - cen) < 1e-12
    z = np.array([-50.0, 0.0, 3.9, 4.0, 4.1, 7.999, 8.0, 8.001, 100.0])
    np.testing.assert_array_equal(ens.select_gp(z), [0, 0, 0, 0, 1, 1, 1, 2, 2])
    rng = np.random.default_rng(0)
    zr = rng.uniform(-5, 15, 500)
    loop = [min(range(3), key=lambda k: (abs(v - ens.centroids[k, 0]), k)) for v in zr]       # scalar restatement
    np.testing.assert_array_equal(ens.select_gp(zr), loop)
    x = rng.standard_normal((4, 7)); u = rng.standard_normal((4, 2))
    np.testing.assert_array_equal(ens.get_z(x, u), x[:, 3])
    # different centroids in the two dimensions: not usable by the reference's MPC path, rejected here
    other = _ensemble_models(tmp_path, centroids=(6.0, 2.0, 11.0), dims=(4,))
    with pytest.raises(ValueError):
        gp_loader.GPEnsemble.from_pickled([m for m in saved if m["reg_dim"] == 3] + other)
    with pytest.raises(ValueError):                              # the plain (one model per dimension) path still refuses ensembles
        gp_loader.gps_from_pickled(saved[:4])


def test_multi_feature_regressor_round_trip_and_oracle_jacobian(tmp_path, oracle):
    """A regressor over (v_x, delta, a) with one length scale per feature (the reference's anisotropic kernel, gp.py:81-138): file
    round trip, predictive mean inside the oracle's dynamics against a direct numpy evaluation, and the oracle's analytic Jacobian
    (all three features, state and input columns) against central differences."""
    rng = np.random.default_rng(3)
    Z = np.c_[rng.uniform(2, 12, 24), rng.uniform(-0.3, 0.3, 24), rng.uniform(-3, 3, 24)]
    e = dict(feat=[3, 6, 7], out=4, Z=Z, alpha=0.2 * rng.standard_normal(24), length_scale=[2.0, 0.2, 1.5], sigma_f=0.7, ymean=0.05)
    d = gp_loader.save_regressor(str(tmp_path / "mf.pkl"), e)
    assert d["x_features"] == [3, 6] and d["u_features"] == [0] and d["x_train"].shape == (24, 3)
    g = gp_loader.gp_entry_from_saved(d)
    assert g["feat"] == [3, 6, 7]
    cfg = default_config(N=20)
    from ad_mpc_amd.config import set_gp
    set_gp(cfg, [g])
    base = default_config(N=20)
    x = np.array([1.0, -2.0, 0.3, 6.0, 0.2, -0.1, 0.05]); u = np.array([0.5, -0.2])
    z = np.array([x[3], x[6], u[0]])
    k = 0.7 * np.exp(-0.5 * (((z - Z) / np.array([2.0, 0.2, 1.5])) ** 2).sum(1))
    mu = float(k @ e["alpha"] + 0.05)
    f0 = oracle.f(base, x, u, 0.4); f1 = oracle.f(cfg, x, u, 0.4)
    assert abs((f1[4] - f0[4]) - mu) <= 1e-13 and np.abs(np.delete(f1 - f0, 4)).max() == 0.0
    Jx, Ju = oracle.jac(cfg, x, u, 0.4)
    h = 1e-6
    for i in range(7):
        dx = np.zeros(7); dx[i] = h
        np.testing.assert_allclose(Jx[:, i], (oracle.f(cfg, x + dx, u, 0.4) - oracle.f(cfg, x - dx, u, 0.4)) / (2 * h), atol=2e-7)
    for j in range(2):
        du = np.zeros(2); du[j] = h
        np.testing.assert_allclose(Ju[:, j], (oracle.f(cfg, x, u + du, 0.4) - oracle.f(cfg, x, u - du, 0.4)) / (2 * h), atol=2e-7)
