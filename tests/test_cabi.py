"""The C-ABI library loads on a machine without a GPU and exports every symbol include/admpc.h declares;
its host-side entry points agree with the Python mirror.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from ad_mpc_amd import _lib
from ad_mpc_amd.config import AdmpcConfig, default_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "admpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(admpc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    qh = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "admpc_quad.h")).read(), flags=re.S)
    qdecl = set(re.findall(r"\b(admpc_quad_[a-z0-9_]+)\s*\(", qh))
    assert qdecl == set(_lib.QUAD_EXPORTS), qdecl ^ set(_lib.QUAD_EXPORTS)
    for name in qdecl:
        assert getattr(lib, name) is not None


def test_quad_default_config_matches_python_mirror(lib):
    from ad_mpc_amd.quad_config import AdmpcQuadConfig, default_quad_config
    c = AdmpcQuadConfig()
    lib.admpc_quad_default_config(C.byref(c))
    assert bytes(c) == bytes(default_quad_config()), "struct layout or default values differ between include/admpc_quad.h and ad_mpc_amd/quad_config.py"


def test_default_config_matches_python_mirror(lib):
    c = AdmpcConfig()
    assert lib.admpc_default_config(C.byref(c), 20, 0.05) == 0
    py = default_config(N=20, Ts=0.05)
    assert bytes(c) == bytes(py), "struct layout or default values differ between include/admpc.h and ad_mpc_amd/config.py"
    assert lib.admpc_default_config(C.byref(c), 1, 0.05) < 0 and b"bad N" in lib.admpc_last_error()
    assert lib.admpc_default_config(C.byref(c), 500, 0.05) < 0


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.admpc_version()


def test_no_cpu_fallback_in_the_product_path():
    """The product package must not reference the oracle anywhere."""
    pkg = os.path.join(ROOT, "ad_mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ad_mpc_amd.engine import BatchSolver
    with pytest.raises(_lib.AdmpcError):
        BatchSolver(default_config())
