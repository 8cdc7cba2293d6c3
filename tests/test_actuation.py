"""SURVEY 8f-1 / 8f-2 rows around the solve: resample_vel in front of it, the safety / actuation branch behind it.
CPU: the oracle restatement (oracle/actuation_oracle.py) and the product's host mirror (ad_mpc_amd/host.py) against hand-derived
fixtures (tests/golden/actuation.json).  GPU: the kernels against the oracle, exactly."""
import json
import os

import numpy as np
import pytest

from ad_mpc_amd import host
from ad_mpc_amd.config import default_config

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "actuation.json")))


def test_resample_vel_oracle_and_host_mirror_match_hand_derived_cases():
    from oracle import actuation_oracle as ao
    for c in GOLD["resample_vel"]:
        assert ao.resample_vel(c["vel_ref"], c["vx"], c["vy"], c["acc_max"], c["dt"]) == c["expected"], c["why"]
        assert host.resample_vel(c["vel_ref"], c["vx"], c["vy"], c["acc_max"], c["dt"]) == c["expected"], c["why"]


def test_actuation_oracle_and_host_mirror_match_hand_derived_cases():
    from oracle import actuation_oracle as ao
    for c in GOLD["actuation"]:
        x, ref, w = np.array(c["x_opt"]), np.array(c["ref"]), np.array(c["w_opt"])
        cnt, mode, rec, healthy = ao.actuation(c["status"], x, w, ref, c["steering"], c["safe_count"], c["threshold"])
        e = c["expected"]
        assert (cnt, mode, list(rec), healthy) == (e["safe_count"], e["mode"], e["record"], e["healthy"]), c["why"]
        assert host.is_valid_command(x, ref) == e["healthy"]
        cnt2, mode2, rec2 = host.actuation(c["status"], host.is_valid_command(x, ref), c["safe_count"], c["threshold"], host.ackermann_fields(x, w),
                                           c["steering"], -3.0, 3.0, -0.52, 0.52)
        assert (cnt2, mode2, list(rec2)) == (e["safe_count"], e["mode"], e["record"]), c["why"]


@pytest.mark.gpu
def test_resample_vel_kernel_exact(gpu_engine_factory):
    import torch
    from oracle import actuation_oracle as ao
    eng = gpu_engine_factory(default_config())
    rng = np.random.default_rng(5)
    B, H = 300, 21
    vx = rng.uniform(0, 15, B); vy = rng.uniform(-1, 1, B); vref = rng.uniform(0, 20, (B, H))
    for c in GOLD["resample_vel"]:                                    # the fixtures first (padded into rows of H)
        pass
    full = rng.uniform(0, 20, (B, 6, H)); full[:, 3, :] = vref           # as row 3 of admpc_waypoints_batch's out_ref (ld = 6 H)
    t = eng.to_device(full)
    eng.resample_vel(t[:, 3, :], eng.to_device(vx), eng.to_device(vy), 5.0, 0.05)
    torch.cuda.synchronize()
    out = t.cpu().numpy()
    exp = np.array([ao.resample_vel(vref[b], vx[b], vy[b], 5.0, 0.05) for b in range(B)])
    np.testing.assert_array_equal(out[:, 3, :], exp)
    others = [0, 1, 2, 4, 5]
    np.testing.assert_array_equal(out[:, others, :], full[:, others, :])      # the other rows are untouched
    for c in GOLD["resample_vel"]:
        v = eng.to_device(np.array([c["vel_ref"]]))
        eng.resample_vel(v, eng.to_device(np.array([c["vx"]])), eng.to_device(np.array([c["vy"]])), c["acc_max"], c["dt"])
        assert v.cpu().numpy()[0].tolist() == c["expected"], c["why"]


@pytest.mark.gpu
def test_actuation_kernel_exact_and_argmin_over_valid_candidates(gpu_engine_factory):
    import torch
    from oracle import actuation_oracle as ao
    from ad_mpc_amd.scenarios import random_scenarios
    cfg = default_config()
    N = cfg.N
    eng = gpu_engine_factory(cfg)
    s = random_scenarios(96, seed=5)
    x, u, cost, st, it = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    x[3, :, 1] += 30.0; x[7, 4, 0] += 30.0                             # two unhealthy predictions
    st = st.copy(); st[[5, 11]] = 4                                       # two solver failures
    u[20, 0, 1] = 7.5; u[21, 0, 1] = -9.0                                 # rate clips
    rng = np.random.default_rng(2)
    steer = rng.uniform(-0.5, 0.5, 96)
    cnt = rng.integers(0, 20, 96).astype(np.int32); cnt[:4] = 9; cnt[3] = 30; cnt[[20, 21]] = 15
    ref = s["xref"][:, :, :2].copy()
    tcnt = torch.as_tensor(cnt, device=eng.device)
    tcost = eng.to_device(cost)
    ack, mode, valid = eng.actuation(eng.to_device(x), eng.to_device(u), eng.to_device(ref), torch.as_tensor(st, device=eng.device),
                                     eng.to_device(steer), tcnt, threshold=10, cost=tcost)
    torch.cuda.synchronize()
    ack, mode, valid, cnt_out, cost_out = ack.cpu().numpy(), mode.cpu().numpy(), valid.cpu().numpy(), tcnt.cpu().numpy(), tcost.cpu().numpy()
    for b in range(96):
        c2, m2, rec, healthy = ao.actuation(int(st[b]), x[b], u[b].reshape(-1), s["xref"][b], float(steer[b]), int(cnt[b]), 10)
        assert (int(cnt_out[b]), int(mode[b]), bool(valid[b])) == (c2, m2, healthy), b
        np.testing.assert_array_equal(ack[b], np.array(rec, dtype=np.float32))
        assert cost_out[b] == (cost[b] if m2 else np.inf)
    assert mode[3] == 0 and mode[5] == 0 and mode[20] == 1 and 0 < mode.sum() < 96
    v, i = eng.argmin(tcost)
    torch.cuda.synchronize()
    masked = np.where(mode == 1, cost, np.inf)
    assert i.item() == int(np.argmin(masked)) and mode[i.item()] == 1
    for c in GOLD["actuation"]:                                           # the hand-derived cases through the kernel
        e = c["expected"]
        cfg4 = default_config(N=c["N"]); e4 = gpu_engine_factory(cfg4)
        xx = np.array(c["x_opt"])[None]; ww = np.array(c["w_opt"]).reshape(1, c["N"], 2); rr = np.array(c["ref"])[None, :, :2].copy()
        tc = torch.as_tensor(np.array([c["safe_count"]], dtype=np.int32), device=e4.device)
        a, m, vl = e4.actuation(e4.to_device(xx), e4.to_device(ww), e4.to_device(rr), torch.as_tensor(np.array([c["status"]], dtype=np.int32), device=e4.device),
                                e4.to_device(np.array([c["steering"]])), tc, threshold=c["threshold"])
        torch.cuda.synchronize()
        assert (int(tc.item()), int(m.item()), bool(vl.item())) == (e["safe_count"], e["mode"], e["healthy"]), c["why"]
        np.testing.assert_array_equal(a.cpu().numpy()[0], np.array(e["record"], dtype=np.float32))
