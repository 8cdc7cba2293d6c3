#!/usr/bin/env python3
"""Measurement of the SURVEY 8f "next" rows that are built (one JSON line each; not the headline bench):
  8f-1 reference generator  admpc_waypoints_batch  (poses/s, vs the CPU oracle = the reference's numpy algorithm)
  8f-2 post-solve epilogue  admpc_epilogue_batch   (records/s, vs the host logic in ad_mpc_amd/host.py)
Only scripts/ and tests/ may use oracle/; this is a measurement script, not product code."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.ref_traj import RefTrajectory
from ad_mpc_amd import host
from oracle import ref_traj_oracle


def timed(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    B = 65536
    rng = np.random.default_rng(0)
    # a closed track of 2000 waypoints
    s = np.linspace(0, 2 * np.pi, 2000)
    xr, yr = 200 * np.cos(s), 120 * np.sin(s)
    psi = np.unwrap(np.arctan2(np.gradient(yr), np.gradient(xr)))
    vel = 8 + 2 * np.sin(3 * s)
    rt = RefTrajectory(traj_horizon=20, traj_dt=0.05, device=0)
    rt.set_traj(xr, yr, psi, vel)
    k = rng.integers(0, 2000, B)
    X = torch.tensor(xr[k] + rng.normal(0, 1, B), dtype=torch.float64, device="cuda:0")
    Y = torch.tensor(yr[k] + rng.normal(0, 1, B), dtype=torch.float64, device="cuda:0")
    P = torch.tensor(psi[k] + rng.normal(0, 0.1, B), dtype=torch.float64, device="cuda:0")
    dt_gpu = timed(lambda: rt.get_waypoints_batch(X, Y, P))
    ncpu = 200
    Xh, Yh, Ph = X[:ncpu].cpu().numpy(), Y[:ncpu].cpu().numpy(), P[:ncpu].cpu().numpy()
    t = time.perf_counter()
    for i in range(ncpu):
        ref_traj_oracle.get_waypoints(rt.trajectory, 20, 0.05, float(Xh[i]), float(Yh[i]), float(Ph[i]))
    dt_cpu = (time.perf_counter() - t) / ncpu
    print(json.dumps({"row": "8f-1 reference generator", "metric": "poses/s", "value": B / dt_gpu, "batch": B, "waypoints": 2000, "horizon": 20,
                      "cpu_baseline": {"value": 1.0 / dt_cpu, "kind": "port (numpy restatement of ref_traj.py:89-171, one thread)"}}))

    cfg = default_config(N=20)
    eng = BatchSolver(cfg, device=0)
    Be = 65536
    xopt = torch.randn(Be, 21, 7, dtype=torch.float64, device=eng.device)
    uopt = torch.randn(Be, 20, 2, dtype=torch.float64, device=eng.device)
    xy = torch.randn(Be, 21, 2, dtype=torch.float64, device=eng.device)
    dt_gpu = timed(lambda: eng.epilogue(xopt, uopt, xy))
    xo, uo, xyo = xopt[:500].cpu().numpy(), uopt[:500].cpu().numpy(), xy[:500].cpu().numpy()
    t = time.perf_counter()
    for i in range(500):
        host.is_valid_command(xo[i], xyo[i]); host.ackermann_fields(xo[i], uo[i].reshape(-1))
    dt_cpu = (time.perf_counter() - t) / 500
    print(json.dumps({"row": "8f-2 post-solve epilogue", "metric": "records/s", "value": Be / dt_gpu, "batch": Be,
                      "cpu_baseline": {"value": 1.0 / dt_cpu, "kind": "host logic ad_mpc_amd/host.py (numpy, one thread)"}}))


if __name__ == "__main__":
    main()
