"""The oracle pins parity, so its own hygiene is tested rather than asserted in prose:
  * `make -C oracle asan`: the restatement under AddressSanitizer + UBSan runs the same solves (every horizon class, failing
    instances, SQP mode, shift, QP debug path) without a report and returns what the plain build returns;
  * `make -C oracle longdouble`: the same file in 80-bit arithmetic bounds the rounding error of the fp64 IPM (the number DESIGN.md
    section 9 quotes for the parity tolerances).
CPU only."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios
san, ref = Oracle(variant="asan"), Oracle()
worst = 0.0
for N, B, kw in ((2, 6, {}), (3, 5, {}), (20, 24, {}), (40, 10, {}), (80, 3, {}), (20, 8, dict(sqp_iters=6, sqp_tol=1e-6))):
    cfg = default_config(N=N, **kw)
    s = random_scenarios(B, N=N, seed=5 + N, blend=(3.0, 5.0))
    bad = random_scenarios(2, N=N, seed=3, blend=(3.0, 5.0), init="zeros")
    s = {k: np.concatenate([s[k], bad[k]]) for k in s}
    a = san.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    b = ref.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    assert (a[3] == b[3]).all() and (a[4] == b[4]).all(), (N, a[3], b[3])
    ok = a[3] != 4
    worst = max(worst, float(np.abs(a[1][ok] - b[1][ok]).max()))
    xs, us = san.shift_batch(cfg, a[0][ok], a[1][ok], s["p"][ok])
    xr, ur = ref.shift_batch(cfg, b[0][ok], b[1][ok], s["p"][ok])
    worst = max(worst, float(np.abs(xs - xr).max()))
    d = san.qp_debug(cfg, s["x0"][0], s["yref"][0], s["yref_e"][0], s["p"][0], s["xbar"][0], s["ubar"][0])
    assert np.isfinite(d["du"]).all()
print("WORST", worst)
"""


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(p)


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "oracle"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=24")
    r = subprocess.run([sys.executable, "-c", _CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    worst = float(r.stdout.split("WORST")[1])
    assert worst <= 1e-12                                  # -O1 + sanitizers against -O2: same arithmetic (contraction is off in both)


def test_fp64_ipm_against_80_bit_arithmetic():
    """The fp64 oracle against the same file with every real in x87 extended precision: same statuses, controls within 1e-10
    (measured 3e-12 at N = 20) -- the rounding noise floor the GPU parity tolerances of DESIGN.md section 9 sit above."""
    from oracle.oracle import Oracle
    from ad_mpc_amd.config import default_config
    from ad_mpc_amd.scenarios import random_scenarios
    ld, ref = Oracle(variant="ld"), Oracle()
    for N, B, tol in ((20, 64, 1e-10), (40, 16, 1e-9)):
        cfg = default_config(N=N)
        s = random_scenarios(B, N=N, seed=21, blend=(3.0, 5.0))
        a = ld.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
        b = ref.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
        np.testing.assert_array_equal(a[3], b[3])
        assert (a[3] == 0).all()
        same = a[4] == b[4]                                  # an iteration count may flip where a stop test sits on the threshold
        assert same.mean() >= 0.9
        d = np.abs(a[1][same] - b[1][same]).max()
        print("N=%d  max|du| fp64 vs 80-bit = %.2e  (same iteration count: %d of %d)" % (N, d, same.sum(), B))
        assert d <= tol
