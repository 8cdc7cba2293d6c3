"""N>1 path on CPU: world_size-2 gloo run of the sharding + arg-min exchange.
Each rank evaluates the cost of its shard with the oracle (checker role only) and the collective
must return the same winner as the single-process scan."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ad_mpc_amd import dist as adist
        from ad_mpc_amd.config import default_config
        from ad_mpc_amd.scenarios import random_scenarios
        from oracle.oracle import Oracle
        lo, hi = adist.shard_range(total, rank, world)
        s = random_scenarios(hi - lo, seed=seed, start=lo)
        cfg = default_config()
        _, _, cost, st, _ = Oracle().solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
        cost_t = torch.from_numpy(cost)
        v, i = adist.local_argmin_torch(cost_t, index_offset=lo)
        gv, gi = adist.global_argmin(v, i)
        # tie test: every rank proposes the same value -> lowest global index must win
        tv, ti = adist.global_argmin(torch.tensor([1.5], dtype=torch.float64), torch.tensor([100 - rank], dtype=torch.int64))
        q.put((rank, float(gv), int(gi), float(tv), int(ti), cost.tolist(), lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_argmin_matches_single_process_scan():
    world, total, seed = 2, 48, 1234
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, seed, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    res.sort()
    full = np.concatenate([np.array(r[5]) for r in res])
    assert len(full) == total
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]          # every rank sees the same winner
    assert res[0][2] == int(np.argmin(full)) and res[0][1] == full.min()
    assert res[0][3] == 1.5 and res[0][4] == 99 and res[1][4] == 99   # tie -> lowest global index
