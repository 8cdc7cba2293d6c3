"""The generated assembly blocks of the condensed kernels, executed by a lane interpreter on the CPU (tests/asm_emu.py): the forward /
backward substitutions, the Newton-row build, and their BORDERED variants (admpc_seg.hip: lanes 40.. carry full rows of a border matrix)."""
import importlib.util
import os

import numpy as np
import pytest

from asm_emu import Wave, WAVE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_subst_asm", os.path.join(ROOT, "ad_mpc_amd", "csrc", "gen_subst_asm.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)

N = 40
LP, LB, PUB = 0, 8 * 1000, 8 * 4000           # byte addresses: packed factor rows, border rows [b][40], publish buffer


def _factor(rng):
    L = np.tril(rng.normal(size=(N, N)) * 0.3, -1) + np.eye(N)
    return L


def _load_factor(w, L, C=None):
    for i in range(N):
        for j in range(i + 1):
            w.lds[LP + 8 * (i * (i + 1) // 2 + j)] = 0.0 if i == j else L[i, j]      # diagonal slots hold 0.0
    if C is not None:
        for b in range(C.shape[0]):
            for j in range(N):
                w.lds[LB + 8 * (b * N + j)] = C[b, j]


@pytest.mark.parametrize("nrows", [40, 47, 54])
def test_forward_substitution_with_border_rows(nrows):
    rng = np.random.default_rng(nrows)
    L = _factor(rng); nb = nrows - N
    C = rng.normal(size=(nb, N)) if nb else None
    y = rng.normal(size=WAVE)
    w = Wave(); _load_factor(w, L, C)
    lane = np.arange(WAVE)
    w.v[100] = y.copy()
    w.v[102] = np.where(lane < N, LP + 8 * (lane * (lane + 1) // 2), np.where(lane < nrows, LB + 8 * (lane - N) * N, LP))
    w.v[103] = PUB + 8 * (lane & 15)
    w.run(gen.fwd(N, nrows if nb else None))
    z = np.linalg.solve(L, y[:N])
    assert np.abs(w.v[100][:N] - z).max() < 1e-12
    if nb:
        assert np.abs(w.v[100][N:nrows] - (y[N:nrows] - C @ z)).max() < 1e-12
    assert w.exec.all()


def test_backward_substitution():
    rng = np.random.default_rng(5)
    L = _factor(rng); y = rng.normal(size=WAVE)
    w = Wave(); _load_factor(w, L)
    lane = np.arange(WAVE)
    w.v[100] = y.copy(); w.v[102] = np.where(lane < N, LP + 8 * lane, LP); w.v[103] = PUB + 8 * (lane & 15)
    w.run(gen.bwd(N))
    assert np.abs(w.v[100][:N] - np.linalg.solve(L.T, y[:N])).max() < 1e-12


@pytest.mark.parametrize("nrows", [40, 47, 54])
def test_newton_row_build_with_border_rows(nrows):
    rng = np.random.default_rng(7 + nrows)
    H = rng.normal(size=(N, N)); H = H + H.T
    nb = nrows - N
    C = rng.normal(size=(max(nb, 1), N))
    w = Wave()
    for i in range(N):
        for j in range(i + 1):
            w.lds[LP + 8 * (i * (i + 1) // 2 + j)] = H[i, j]
    for b in range(nb):
        for j in range(N):
            w.lds[LB + 8 * (b * N + j)] = C[b, j]
    lane = np.arange(WAVE)
    dbar = rng.normal(size=WAVE); sodd = np.where(lane < N, rng.normal(size=WAVE), 0.0)
    row = np.zeros((WAVE, N))
    for half, (lo, hi) in enumerate(((0, N // 2), (N // 2, N))):
        cnt = hi - lo
        ops = ["v[%d:%d]" % (2 * q, 2 * q + 1) for q in range(cnt)] + ["v200", "v[202:203]", "v[204:205]"]
        w.v[200] = np.where(lane < N, LP + 8 * (lane * (lane + 1) // 2), np.where(lane < nrows, LB + 8 * (lane - N) * N, LP))
        w.v[202] = dbar.copy(); w.v[204] = sodd.copy()
        w.run(gen.rowbuild(N, lo, hi, nrows if nb else None), ops)
        for q in range(cnt):
            row[:, lo + q] = w.v[2 * q]
    for i in range(N):
        for c in range(N):
            want = (H[i, c] if c <= i else 0.0) + (sodd[i] if c & 1 else 0.0) + (dbar[i] if c == i else 0.0)
            assert abs(row[i, c] - want) < 1e-13, (i, c)
    for b in range(nb):
        assert np.abs(row[N + b] - C[b]).max() < 1e-13       # full rows, no diagonal, no s_odd (the caller passes 0 there)
    assert np.abs(row[nrows:]).max() == 0.0


@pytest.mark.parametrize("nrows", [40, 47, 54])
def test_symmetric_row_with_border_rows(nrows):
    rng = np.random.default_rng(11 + nrows)
    H = rng.normal(size=(N, N)); H = H + H.T
    nb = nrows - N
    C = rng.normal(size=(max(nb, 1), N))
    w = Wave()
    for i in range(N):
        for j in range(i + 1):
            w.lds[LP + 8 * (i * (i + 1) // 2 + j)] = H[i, j]
    for b in range(nb):
        for j in range(N):
            w.lds[LB + 8 * (b * N + j)] = C[b, j]
    lane = np.arange(WAVE)
    row = np.zeros((WAVE, N))
    for (lo, hi) in ((0, N // 2), (N // 2, N)):
        cnt = hi - lo
        ops = ["v[%d:%d]" % (2 * q, 2 * q + 1) for q in range(cnt)] + ["v200", "v201"]
        w.v[200] = np.where(lane < N, LP + 8 * (lane * (lane + 1) // 2), np.where(lane < nrows, LB + 8 * (lane - N) * N, LP))
        w.v[201] = np.where(lane < N, LP + 8 * lane, LP)
        w.run(gen.symrow(N, lo, hi, nrows if nb else None), ops)
        for q in range(cnt):
            row[:, lo + q] = w.v[2 * q]
    assert np.abs(row[:N] - H).max() < 1e-13
    for b in range(nb):
        assert np.abs(row[N + b] - C[b]).max() < 1e-13
    assert np.abs(row[nrows:]).max() == 0.0
