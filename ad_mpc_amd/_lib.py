"""Loader of the C-ABI shared library ``libadmpc.so`` (include/admpc.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the solver
surface raises.  The CPU oracle under ``oracle/`` is test infrastructure and is never used here.
"""
import ctypes as C
import os

from .config import AdmpcConfig
from .quad_config import AdmpcQuadConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadmpc.so")

EXPORTS = (
    "admpc_default_config", "admpc_create", "admpc_destroy", "admpc_reserve", "admpc_solve_batch", "admpc_solve_batch_ex", "admpc_nlp_residuals_batch", "admpc_solve_batch_f32", "admpc_shoot_batch",
    "admpc_argmin", "admpc_argmin_pairs", "admpc_argmin_pairs_host", "admpc_argmin_global", "admpc_select_cluster_batch", "admpc_solve_batch_routed", "admpc_shift_batch", "admpc_epilogue_batch", "admpc_actuation_batch", "admpc_resample_vel_batch", "admpc_waypoints_batch", "admpc_last_error", "admpc_version",
)
QUAD_EXPORTS = ("admpc_quad_default_config", "admpc_quad_create", "admpc_quad_destroy", "admpc_quad_solve_batch", "admpc_quad_solve_batch_ex", "admpc_quad_select_cluster_batch",
                "admpc_quad_solve_batch_routed", "admpc_quad_shoot_batch", "admpc_quad_shoot_batch_ex")   # include/admpc_quad.h

_lib = None


class AdmpcError(RuntimeError):
    pass


def load():
    """dlopen libadmpc.so and declare the prototypes of include/admpc.h (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdmpcError("%s not found: build it with `make -C ad_mpc_amd/csrc` (or __graft_entry__.build()); "
                         "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p     # device pointers travel as integers
    cp = C.POINTER(AdmpcConfig)
    L.admpc_default_config.argtypes = [cp, C.c_int, C.c_double]; L.admpc_default_config.restype = C.c_int
    L.admpc_create.argtypes = [cp, C.c_int, C.POINTER(C.c_void_p)]; L.admpc_create.restype = C.c_int
    L.admpc_destroy.argtypes = [C.c_void_p]; L.admpc_destroy.restype = None
    L.admpc_reserve.argtypes = [C.c_void_p, C.c_int]; L.admpc_reserve.restype = C.c_int
    L.admpc_solve_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, ip, vp]
    L.admpc_solve_batch.restype = C.c_int
    L.admpc_solve_batch_ex.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp, dp, vp]
    L.admpc_solve_batch_ex.restype = C.c_int
    L.admpc_nlp_residuals_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, vp]
    L.admpc_nlp_residuals_batch.restype = C.c_int
    L.admpc_solve_batch_f32.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, ip, vp]
    L.admpc_solve_batch_f32.restype = C.c_int
    L.admpc_shoot_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, vp]; L.admpc_shoot_batch.restype = C.c_int
    L.admpc_argmin.argtypes = [C.c_void_p, dp, C.c_int, C.c_int64, dp, ip, vp]; L.admpc_argmin.restype = C.c_int
    L.admpc_argmin_pairs.argtypes = [C.c_void_p, dp, C.c_int, dp, ip, vp]; L.admpc_argmin_pairs.restype = C.c_int
    L.admpc_argmin_pairs_host.argtypes = [dp, C.c_int, dp, ip]; L.admpc_argmin_pairs_host.restype = C.c_int
    L.admpc_argmin_global.argtypes = [C.c_void_p, dp, C.c_int, C.c_int64, C.c_void_p, dp, ip, vp]; L.admpc_argmin_global.restype = C.c_int
    L.admpc_select_cluster_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), dp, dp, C.c_int, dp, ip, vp]; L.admpc_select_cluster_batch.restype = C.c_int
    L.admpc_solve_batch_routed.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, ip, dp, dp, dp, dp, dp, dp, dp, ip, ip, vp]; L.admpc_solve_batch_routed.restype = C.c_int
    L.admpc_shift_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, C.c_int, vp]; L.admpc_shift_batch.restype = C.c_int
    L.admpc_epilogue_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, ip, vp]; L.admpc_epilogue_batch.restype = C.c_int
    L.admpc_actuation_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, ip, dp, ip, C.c_int, dp, dp, ip, ip, vp]; L.admpc_actuation_batch.restype = C.c_int
    L.admpc_resample_vel_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, dp, vp]; L.admpc_resample_vel_batch.restype = C.c_int
    L.admpc_waypoints_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int] + [dp] * 13 + [vp]
    L.admpc_waypoints_batch.restype = C.c_int
    qp = C.POINTER(AdmpcQuadConfig)
    L.admpc_quad_default_config.argtypes = [qp]; L.admpc_quad_default_config.restype = None
    L.admpc_quad_create.argtypes = [qp, C.c_int, C.POINTER(C.c_void_p)]; L.admpc_quad_create.restype = C.c_int
    L.admpc_quad_destroy.argtypes = [C.c_void_p]; L.admpc_quad_destroy.restype = None
    L.admpc_quad_solve_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, ip, ip, vp]; L.admpc_quad_solve_batch.restype = C.c_int
    L.admpc_quad_shoot_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, vp]; L.admpc_quad_shoot_batch.restype = C.c_int
    L.admpc_quad_solve_batch_ex.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, ip, vp]; L.admpc_quad_solve_batch_ex.restype = C.c_int
    L.admpc_quad_shoot_batch_ex.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, vp]; L.admpc_quad_shoot_batch_ex.restype = C.c_int
    L.admpc_quad_select_cluster_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), dp, dp, C.c_int, dp, ip, vp]; L.admpc_quad_select_cluster_batch.restype = C.c_int
    L.admpc_quad_solve_batch_routed.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, ip, dp, dp, dp, dp, dp, dp, dp, ip, ip, vp]; L.admpc_quad_solve_batch_routed.restype = C.c_int
    L.admpc_last_error.restype = C.c_char_p
    L.admpc_version.restype = C.c_char_p
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise AdmpcError("libadmpc error %d: %s" % (rc, load().admpc_last_error().decode()))
