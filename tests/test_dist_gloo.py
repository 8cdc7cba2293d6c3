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


SHARDS = (3, 6)          # the two gloo ranks play shards 3 and 6 of the 8-way config-4 sharding of 65536 scenarios


def _worker(rank, world, port, per_rank, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ad_mpc_amd import dist as adist
        from ad_mpc_amd.config import default_config
        from ad_mpc_amd.scenarios import random_scenarios
        from oracle.oracle import Oracle
        lo, hi = adist.shard_range(65536, SHARDS[rank], 8)
        assert hi - lo == 8192
        s = random_scenarios(per_rank, seed=seed, start=lo)                 # the first instances of the shard (the oracle is the checker here)
        cfg = default_config()
        _, _, cost, st, _ = Oracle().solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
        cost_t = torch.from_numpy(cost)
        # the device path's record flow: local arg-min -> 16-byte record -> all-gather -> second-level arg-min over the records
        v, i = adist.local_argmin_torch(cost_t, index_offset=lo)
        win = adist.global_argmin_records(adist.pack_pair(v, i), adist.pairs_min_torch)
        gv, gi = adist.unpack_pair(win)
        # ties across ranks: the same value from both -> the lowest GLOBAL index wins; a NaN record never wins
        tie = adist.global_argmin_records(adist.pack_pair(torch.tensor([1.5], dtype=torch.float64), torch.tensor([lo + 5], dtype=torch.int64)), adist.pairs_min_torch)
        nanr = adist.global_argmin_records(adist.pack_pair(torch.tensor([float("nan") if rank == 0 else 2.5], dtype=torch.float64),
                                                           torch.tensor([lo], dtype=torch.int64)), adist.pairs_min_torch)
        # and the older (value, index) interface gives the same winner
        ov, oi = adist.global_argmin(v, i)
        q.put((rank, gv, gi, adist.unpack_pair(tie), adist.unpack_pair(nanr), cost.tolist(), lo, float(ov), int(oi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_record_path_matches_single_process_scan():
    world, per_rank, seed = 2, 24, 1234
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, seed, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    res.sort()
    los = [r[6] for r in res]
    assert los == [3 * 8192, 6 * 8192]
    costs = [np.array(r[5]) for r in res]
    best_rank = int(np.argmin([c.min() for c in costs]))
    expect = (float(costs[best_rank].min()), los[best_rank] + int(np.argmin(costs[best_rank])))
    for r in res:
        assert (r[1], r[2]) == expect                                     # every rank sees the same winner, with its GLOBAL index
        assert r[3] == (1.5, los[0] + 5)                                  # tie -> lowest global index (rank 0's shard)
        assert r[4] == (2.5, los[1])                                      # NaN never wins
        assert (r[7], r[8]) == expect
