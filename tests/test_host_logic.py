"""Host orchestration pieces (H7/H8) against hand-derived I/O pairs and a literal loop-form restatement
of the reference lines (ad_3d_optimizer.py:347-349,385-394,423-437,443,475; create_ros_ad_mpc.py:95-98)."""
import math

import numpy as np
import pytest

from ad_mpc_amd import host
from ad_mpc_amd.scenarios import random_scenarios, assemble, straight_scenario
from ad_mpc_amd.dist import shard_range, pick_min, local_argmin_torch


def test_yaw_fix_hand_cases():
    pi = math.pi
    # psi0 < 0 and psi0 + pi < ref  -> ref - 2pi
    assert host.yaw_fix(-3.0, 3.0) == pytest.approx(3.0 - 2 * pi)
    assert host.yaw_fix(-3.0, 0.1) == pytest.approx(0.1)                  # -3+pi = 0.1416 > 0.1 -> unchanged
    # psi0 > 0 and psi0 - pi > ref  -> ref + 2pi
    assert host.yaw_fix(3.0, -3.0) == pytest.approx(-3.0 + 2 * pi)
    assert host.yaw_fix(3.0, -0.1) == pytest.approx(-0.1)                 # 3-pi = -0.1416 < -0.1 -> unchanged
    # psi0 == 0 -> never changed
    assert host.yaw_fix(0.0, 5.0) == 5.0 and host.yaw_fix(0.0, -5.0) == -5.0


def _yaw_fix_loop(psi0, ref):
    out = []
    for r in ref:
        if psi0 < 0:
            if psi0 + math.pi < r:
                r = r - 2 * math.pi
        elif psi0 > 0:
            if psi0 - math.pi > r:
                r = r + 2 * math.pi
        out.append(r)
    return np.array(out)


def test_yaw_fix_vectorised_equals_loop_form():
    rng = np.random.default_rng(0)
    for _ in range(200):
        psi0 = rng.uniform(-math.pi, math.pi); ref = rng.uniform(-4, 4, 21)
        np.testing.assert_array_equal(host.yaw_fix(psi0, ref), _yaw_fix_loop(psi0, ref))


def test_vel_switch():
    assert host.vel_switch(99.0, 100.0, 110.0) == 0.0
    assert host.vel_switch(105.0, 100.0, 110.0) == pytest.approx(0.5)
    assert host.vel_switch(500.0, 100.0, 110.0) == 1.0
    assert host.vel_switch(4.0, 3.0, 5.0) == pytest.approx(0.5)


def test_pad_reference_repeats_last_row():
    x = np.arange(3 * 7, dtype=float).reshape(3, 7); u = np.arange(2 * 2, dtype=float).reshape(2, 2)
    xp, up = host.pad_reference(x, u, N=5)
    assert xp.shape == (6, 7) and up.shape == (5, 2)          # one u row appended per appended x row
    assert (xp[3:] == x[-1]).all() and (up[2:] == u[-1]).all()
    assert xp is not x and (x == np.arange(21).reshape(3, 7)).all()


def test_is_valid_command_thresholds():
    N = 20
    ref = np.zeros((N + 1, 7)); x = np.zeros((N + 1, 7))
    assert host.is_valid_command(x, ref)
    x2 = x.copy(); x2[:, 1] = 2.9                              # mean 2.9*20/21 = 2.76 < 3, max 2.9 < 4, var small
    assert host.is_valid_command(x2, ref)
    x3 = x.copy(); x3[:, 1] = 3.2                              # mean 3.05 >= 3
    assert not host.is_valid_command(x3, ref)
    x4 = x.copy(); x4[5, 0] = 4.1                              # max >= 4
    assert not host.is_valid_command(x4, ref)
    x5 = x.copy(); x5[:10, 1] = 3.9                            # unbiased variance of (10x3.9, 11x0) = 3.98 >= 2
    assert not host.is_valid_command(x5, ref)
    x6 = x.copy(); x6[N, 0] = 100.0                            # the last slot is never measured (loop stops at len-1)
    assert host.is_valid_command(x6, ref)


def test_fallback_command_is_the_references_2n_minus_1_slice():
    prev = np.arange(40.0)
    w = host.fallback_command(prev)
    assert w.shape == (39,)
    np.testing.assert_array_equal(w[:37], prev[2:39]); np.testing.assert_array_equal(w[37:], prev[37:39])


def test_ackermann_mapping():
    x = np.zeros((21, 7)); x[0, 6] = 0.12; x[0, 3] = 7.5
    w = np.array([1.5, -0.3] + [0.0] * 38)
    assert host.ackermann_fields(x, w) == (0.12, -0.3, 7.5, 1.5)


def test_scenarios_are_shard_invariant_and_well_formed():
    full = random_scenarios(16, seed=1234)
    lo, hi = shard_range(16, 1, 2)
    part = random_scenarios(hi - lo, seed=1234, start=lo)
    for k in ("x0", "yref", "yref_e", "p", "xbar", "ubar"):
        np.testing.assert_array_equal(full[k][lo:hi], part[k])
    assert full["yref"].shape == (16, 20, 9) and full["xbar"].shape == (16, 21, 7)
    assert (full["p"] == 0).all()                              # shipped blend speeds 100/110 m/s -> kinematic
    assert (random_scenarios(16, blend=(3.0, 5.0))["p"] > 0).any()
    # the yaw reference stays within pi of the initial yaw after the fix whenever the fix applies
    assert np.abs(full["yref"][:, 0, 2] - full["x0"][:, 2]).max() < math.pi


def test_shard_range_covers_everything_once():
    for total, world in [(65536, 8), (10, 3), (7, 8), (0, 2)]:
        got = []
        for r in range(world):
            lo, hi = shard_range(total, r, world); got += list(range(lo, hi))
        assert got == list(range(total))


def test_pick_min_tie_break_and_nan():
    import torch
    v = torch.tensor([3.0, 1.0, float("nan"), 1.0, float("inf")], dtype=torch.float64)
    i = torch.tensor([10, 7, 2, 5, 1], dtype=torch.int64)
    m, k = pick_min(v, i)
    assert m.item() == 1.0 and k.item() == 5                    # lowest index among the tied minima, NaN ignored
    m, k = local_argmin_torch(torch.tensor([float("inf")] * 3, dtype=torch.float64), 100)
    assert math.isinf(m.item()) and k.item() == 100


def test_assemble_matches_run_optimization_steps():
    x0, xref, uref = straight_scenario()
    x0 = x0.copy(); x0[2] = -3.0; xref = xref.copy(); xref[:, 2] = 3.0
    s = assemble(x0[None], xref[None], uref[None], init="zeros")
    assert s["yref"][0, :, 2] == pytest.approx(3.0 - 2 * math.pi) and s["yref_e"][0, 2] == pytest.approx(3.0 - 2 * math.pi)
    assert (s["yref"][0, :, 7:] == 0).all() and (s["xbar"] == 0).all()
