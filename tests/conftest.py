import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def oracle_omp():
    """The same oracle built with OpenMP over instances (full-size batches stay fast); identical arithmetic per instance."""
    from oracle.oracle import Oracle, build
    build(omp=True)
    return Oracle(omp=True)


@pytest.fixture(scope="session")
def golden_shooting():
    with open(os.path.join(GOLDEN, "shooting.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_kat():
    with open(os.path.join(GOLDEN, "kat_sim_car_iterate.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_kat_mid_rti():
    """The reference's second stored iterate (solve_iteration.json: dynamic branch, in the middle of an RTI sequence); what it pins
    and to which level: oracle/make_golden.py:kat_mid_rti."""
    with open(os.path.join(GOLDEN, "kat_solve_iteration.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_engine_factory():
    """Returns a function cfg -> BatchSolver on cuda:0; imports the HIP path lazily so that the CPU
    suite never touches it."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ad_mpc_amd.engine import BatchSolver
    cache = {}

    def make(cfg):
        return BatchSolver(cfg, device=0)
    return make
