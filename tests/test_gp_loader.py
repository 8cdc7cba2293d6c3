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
    two = dict(d); two["x_features"] = [3, 4]
    with pytest.raises(ValueError):
        gp_loader.gp_entry_from_saved(two)
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
