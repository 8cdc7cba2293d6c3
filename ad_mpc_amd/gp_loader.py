"""Loader of the reference's fitted GP models.

On-disk format (reference ``src/model_fitting/gp.py:489-516``, ``GPRegressor.save``): one ``joblib`` pickle per regressor
holding a dict with the keys ``kernel_params`` ({'l', 'sigma_f'}), ``kernel_type`` ('squared_exponential'), ``x_train``,
``y_train``, ``k_inv_y``, ``k_inv``, ``sigma_n``, ``reg_dim``, ``x_features``, ``u_features``, ``mean``, ``y_mean``.
Directory scheme (``src/utils/utils.py:175-236``): ``<save_dir>/<git>/<model_name>/<file_name>*.pkl`` with
``file_name = '__'.join(('' if v else 'no_') + k for k, v in sorted(params.items()))``.

The engine evaluates the GP mean  mu(z) = sum_i k(z, x_train_i) k_inv_y_i + y_mean  with  k = sigma_f exp(-|z - x|^2 / (2 l^2))
(sigma_f is NOT squared in the reference, ``gp.py:81-138``; ``l`` is a scalar or one length scale per feature) inside the dynamics
(``ad_mpc_amd.config.set_gp``).  Device limits: 1 to 3 features per regressor out of (v_x, v_y, psi_dot, delta, a, delta_dot), at
most 4 regressors and 32 training points each; anything else raises ``ValueError``.

Clustered ensembles (``GPEnsemble``, ``gp.py:536-607``; selection ``select_gp``, ``gp.py:738-770``): the reference keeps one
acados solver per cluster (``quad_3d_optimizer.py:207``) and picks one per solve from the reference state (``:452``, ``:491``).
``GPEnsemble`` below is the host-side restatement (grouping by output dimension, clusters sorted by the first centroid
coordinate, nearest-centroid selection); ``ad_mpc_amd.engine.EnsembleBatchSolver`` holds one engine handle per cluster and
routes every instance of a batch to the handle of its cluster.
"""
import os

import numpy as np

from .config import GP_MAX, GP_MAX_FEAT, GP_MAX_POINTS, NX, NU, set_gp

SAVED_KEYS = ("kernel_params", "kernel_type", "x_train", "y_train", "k_inv_y", "k_inv", "sigma_n", "reg_dim", "x_features",
              "u_features", "mean", "y_mean")


def get_model_dir_and_file(model_options, save_dir):
    """utils.py:175-188 with the save directory passed in instead of read from ``GPConfig.SAVE_DIR``."""
    directory = os.path.join(save_dir, str(model_options["git"]), str(model_options["model_name"]))
    params = model_options["params"]
    parts = [("" if params[k] else "no_") + k for k in sorted(params.keys())]
    return directory, "__".join(parts)


def load_pickled_models(directory="", file_name="", model_options=None, save_dir=None):
    """utils.py:191-236: ``{"models": [dict, ...]}`` for every ``<file_name>*.pkl`` in ``directory`` (``feats.csv`` is skipped
    as in the reference), ``None`` when the directory or a matching file is missing."""
    import joblib
    if model_options is not None:
        directory, file_name = get_model_dir_and_file(model_options, save_dir if save_dir is not None else "")
    try:
        files = sorted(os.listdir(directory))
    except FileNotFoundError:
        return None
    models = []
    for f in files:
        if not f.startswith(file_name) and f != "feats.csv":
            continue
        if ".pkl" not in f:
            continue
        path = os.path.join(directory, f)
        if os.path.isdir(path):
            raise FileNotFoundError("Tried to load file from directory %s, but it was not found." % directory)
        models.append(joblib.load(path))
    return {"models": models} if models else None


def gp_entry_from_saved(d):
    """One saved regressor dict -> the dict ``set_gp`` takes (feat, out, Z, alpha, length_scale, sigma_f, ymean)."""
    missing = [k for k in ("kernel_params", "kernel_type", "x_train", "k_inv_y", "reg_dim", "x_features", "u_features") if k not in d]
    if missing:
        raise ValueError("not a saved GPRegressor: missing %s" % missing)
    if d["kernel_type"] != "squared_exponential":
        raise ValueError("only the squared_exponential kernel is supported (gp.py:60)")
    xf = [int(i) for i in np.atleast_1d(d["x_features"]).reshape(-1)] if d["x_features"] is not None else []
    uf = [int(i) for i in np.atleast_1d(d["u_features"]).reshape(-1)] if d["u_features"] is not None else []
    feats = xf + [NX + i for i in uf]                       # z = [x[x_features]; u[u_features]] (gp.py:609-630)
    if not (1 <= len(feats) <= GP_MAX_FEAT):
        raise ValueError("device GPs take 1..%d features; this regressor has x_features=%s u_features=%s" % (GP_MAX_FEAT, xf, uf))
    if any(not (3 <= f < NX + NU) for f in feats):
        raise ValueError("features must be among v_x, v_y, psi_dot, delta, a, delta_dot (indices 3..8 of [x(7); u(2)]), got %s" % feats)
    Z = np.asarray(d["x_train"], dtype=np.float64)
    Z = Z.reshape(Z.shape[0], -1)
    if Z.shape[1] != len(feats):
        raise ValueError("x_train must be n x %d for this regressor, got %s" % (len(feats), Z.shape,))
    alpha = np.asarray(d["k_inv_y"], dtype=np.float64).reshape(-1)
    if alpha.size != Z.shape[0]:
        raise ValueError("k_inv_y has %d entries for %d training points" % (alpha.size, Z.shape[0]))
    if Z.shape[0] > GP_MAX_POINTS:
        raise ValueError("%d training points; the device holds at most %d" % (Z.shape[0], GP_MAX_POINTS))
    kp = d["kernel_params"]
    ell = np.atleast_1d(np.squeeze(np.asarray(kp["l"], dtype=np.float64))).reshape(-1) if "l" in kp else np.ones(1)
    if ell.size not in (1, len(feats)):
        raise ValueError("anisotropic kernel must have one length scale per feature (gp.py:73-80)")
    ell = float(ell[0]) if ell.size == 1 else ell
    out = int(np.squeeze(d["reg_dim"]))
    if not (0 <= out < NX):
        raise ValueError("reg_dim %d outside the state" % out)
    ymean = float(np.squeeze(d["y_mean"])) if d.get("y_mean") is not None else 0.0
    one = len(feats) == 1
    return dict(feat=feats[0] if one else feats, out=out, Z=Z[:, 0] if one else Z, alpha=alpha, length_scale=ell,
                sigma_f=float(kp.get("sigma_f", 1.0)), ymean=ymean)


class GPEnsemble:
    """Homogeneous clustered ensemble: for every output dimension the same K centroids (what the reference's MPC path can use:
    ``gp_ind`` is ONE key into its dictionary of solvers).  ``clusters[c]`` is the ``set_gp`` list of cluster c (one regressor
    per output dimension), ``centroids`` is K x d (d = 1 for the device's one-feature regressors)."""

    def __init__(self, clusters, centroids, feat):
        self.clusters = clusters
        self.centroids = np.asarray(centroids, dtype=np.float64).reshape(len(clusters), -1)
        self.feats = [int(f) for f in np.atleast_1d(feat).reshape(-1)]      # indices into [x(7); u(2)] of the features the centroids live in
        self.feat = self.feats[0]

    @property
    def n_models(self):
        return len(self.clusters)

    @classmethod
    def from_pickled(cls, pre_trained_models):
        """gp.py:575-607 (add_model per output dimension): group the saved regressors by ``reg_dim``, sort the clusters of a
        dimension by the first coordinate of their centroid (the regressor's ``mean``), require the same centroids in every
        dimension (``homogeneous_feature_space``, gp.py:772-788)."""
        models = pre_trained_models["models"] if isinstance(pre_trained_models, dict) else list(pre_trained_models)
        by_dim = {}
        for m in models:
            e = gp_entry_from_saved(m)
            c = np.atleast_1d(np.asarray(m.get("mean", 0.0), dtype=np.float64)).reshape(-1)
            by_dim.setdefault(e["out"], []).append((c, e))
        if not by_dim:
            raise ValueError("no regressors")
        if len(by_dim) > GP_MAX:
            raise ValueError("%d output dimensions; the device holds at most %d regressors" % (len(by_dim), GP_MAX))
        dims = sorted(by_dim)
        ref_c = None
        feat = None
        for d in dims:
            grp = sorted(by_dim[d], key=lambda ce: ce[0][0])          # argsort of centroids[:, 0] (gp.py:594)
            by_dim[d] = grp
            cen = np.array([c for c, _ in grp])
            if ref_c is None:
                ref_c, feat = cen, grp[0][1]["feat"]
            elif cen.shape != ref_c.shape or np.any(cen != ref_c):
                raise ValueError("non-homogeneous ensemble (different centroids per output dimension): the reference's MPC path "
                                 "cannot use it either (one solver key per solve, quad_3d_optimizer.py:452-462)")
            if any(e["feat"] != grp[0][1]["feat"] for _, e in grp):
                raise ValueError("the clusters of one output dimension must share their feature")
        K = ref_c.shape[0]
        clusters = [[by_dim[d][c][1] for d in dims] for c in range(K)]
        return cls(clusters, ref_c, feat)

    def get_z(self, x, u):
        """gp.py:609-630 for numpy inputs: the feature the selection is made on.  x: (..., 7), u: (..., 2)."""
        x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
        xu = np.concatenate([x, u], axis=-1)
        z = xu[..., self.feats]
        return z[..., 0] if len(self.feats) == 1 else np.moveaxis(z, -1, 0)          # (n,) or (d, n) as select_gp takes it

    def select_gp(self, z):
        """gp.py:738-770: index of the nearest centroid (Euclidean) for every sample; z: (n,) or (d, n).  Ties go to the lowest
        index (numpy.argmin), i.e. to the cluster with the smaller first centroid coordinate."""
        z = np.atleast_2d(np.asarray(z, dtype=np.float64))
        return np.argmin(np.sqrt(np.sum((z[np.newaxis, :, :] - self.centroids[:, :, np.newaxis]) ** 2, 1)), 0)


def gps_from_pickled(pre_trained_models):
    """``{"models": [...]}`` (or a plain list of saved dicts) -> list for ``set_gp``.  Several regressors for the same output
    dimension are the reference's per-cluster ensembles: rejected."""
    models = pre_trained_models["models"] if isinstance(pre_trained_models, dict) else list(pre_trained_models)
    if len(models) > GP_MAX:
        raise ValueError("%d regressors; the device holds at most %d" % (len(models), GP_MAX))
    gps = [gp_entry_from_saved(m) for m in models]
    outs = [g["out"] for g in gps]
    if len(set(outs)) != len(outs):
        raise ValueError("several regressors for one output dimension: a clustered ensemble -- use GPEnsemble.from_pickled "
                         "and engine.EnsembleBatchSolver (one handle per cluster)")
    return sorted(gps, key=lambda g: g["out"])


def install_from_directory(cfg, directory="", file_name="", model_options=None, save_dir=None):
    """Load the pickles and install them into ``cfg``; returns the number of regressors (0 when nothing was found)."""
    pm = load_pickled_models(directory, file_name, model_options, save_dir)
    if pm is None:
        return 0
    gps = gps_from_pickled(pm)
    set_gp(cfg, gps)
    return len(gps)


def save_regressor(path, entry, sigma_n=1e-3):
    """Write one regressor in the reference's format (used by tests and for hand-made models): ``entry`` as for ``set_gp``
    plus optional ``y_train``.  The key set and value shapes follow ``GPRegressor.save`` (gp.py:495-508)."""
    import joblib
    feats = [int(f) for f in np.atleast_1d(entry["feat"]).reshape(-1)]
    Z = np.asarray(entry["Z"], dtype=np.float64).reshape(-1, len(feats))
    d = {
        "kernel_params": {"l": np.atleast_1d(np.asarray(entry["length_scale"], dtype=np.float64)).reshape(-1), "sigma_f": float(entry.get("sigma_f", 1.0))},
        "kernel_type": "squared_exponential",
        "x_train": Z,
        "y_train": np.asarray(entry.get("y_train", np.zeros(Z.shape[0])), dtype=np.float64).reshape(-1, 1),
        "k_inv_y": np.asarray(entry["alpha"], dtype=np.float64).reshape(-1, 1),
        "k_inv": np.eye(Z.shape[0]),
        "sigma_n": float(sigma_n),
        "reg_dim": int(entry["out"]),
        "x_features": [f for f in feats if f < NX],
        "u_features": [f - NX for f in feats if f >= NX],
        "mean": np.atleast_1d(np.asarray(entry.get("centroid", 0.0), dtype=np.float64)),      # the cluster centroid (gp.py:593)
        "y_mean": np.array(float(entry.get("ymean", 0.0))),
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        joblib.dump(d, f)
    return d
