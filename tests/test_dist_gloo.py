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
        # the device path's record flow: local arg-min -> 16-byte record -> all-gather -> second-level arg-min over the records,
        # reduced by libadmpc's admpc_argmin_pairs_host: the source the device kernel is compiled from (csrc/argmin_rule.h)
        v, i = adist.local_argmin_torch(cost_t, index_offset=lo)
        win = adist.global_argmin_records(adist.pack_pair(v, i), adist.pairs_min_host)
        gv, gi = adist.unpack_pair(win)
        # ties across ranks: the same value from both -> the lowest GLOBAL index wins; a NaN record never wins
        tie = adist.global_argmin_records(adist.pack_pair(torch.tensor([1.5], dtype=torch.float64), torch.tensor([lo + 5], dtype=torch.int64)), adist.pairs_min_host)
        nanr = adist.global_argmin_records(adist.pack_pair(torch.tensor([float("nan") if rank == 0 else 2.5], dtype=torch.float64),
                                                           torch.tensor([lo], dtype=torch.int64)), adist.pairs_min_host)
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


def _records(recs):
    t = torch.empty((len(recs), 2), dtype=torch.float64)
    for k, (c, i) in enumerate(recs):
        t[k, 0] = c
        t[k, 1:2].view(torch.int64)[0] = i
    return t


def test_argmin_rules_table_on_the_host_reducers():
    """tests/argmin_spec.py: one table for every implementation of the arg-min rules.  Here: the library's host reducer
    (admpc_argmin_pairs_host -- csrc/argmin_rule.h, the source of the device kernels) and the torch restatement; the device
    kernels run the same table in tests/test_gpu_parity.py."""
    sys.path.insert(0, ROOT)
    from ad_mpc_amd import dist as adist
    from tests import argmin_spec as spec
    assert len(spec.CASES) >= 20
    for name, recs, want in spec.CASES:
        assert spec.reference(recs) == want, name
        for impl in (adist.pairs_min_host, adist.pairs_min_torch):
            got = adist.unpack_pair(impl(_records(recs)))
            assert got[1] == want[1] and (got[0] == want[0]), (name, impl.__name__, got, want)
    for name, costs, off, want in spec.ARRAY_CASES:
        v, i = adist.local_argmin_torch(torch.tensor(costs, dtype=torch.float64), index_offset=off)
        assert (float(v), int(i)) == want, (name, float(v), int(i), want)
        # the first level followed by a one-record second level is the identity
        assert adist.unpack_pair(adist.pairs_min_host(adist.pack_pair(v, i).reshape(1, 2))) == want, name


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks_and_never_reports_fewer():
    """`python bench.py --gpus 2` with no torch.distributed environment must start two ranks itself (it used to
    run one rank and print n_gpus 1).  --dry-collective gloo runs that launcher, the rendezvous, the shard offsets and the record
    path of the arg-min on CPU ranks; the winner must be the single-process arg-min and sit in rank 1's shard.  Without the dry
    switch and without GPUs the bench refuses instead of reporting a smaller job."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-collective", "gloo", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=280)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1                                               # ONE JSON line, relayed from rank 0
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["ranks_seen"] == 2 and js["dry"] is True and js["value"] is None and js["steps"] == 3
    assert js["argmin"] == js["argmin_single_process"] and js["argmin"]["index"] == 10945 >= 8192
    # the inputs of a scaling curve: every N > 1 line carries a per-rank rate measured WITHOUT the collective in the
    # same run at the same per-GPU batch (min / max over ranks, rank 0's as the one-GPU value at this batch), the committed PMC traffic of
    # the 8192-instance shard (no live passes in a multi-rank run), and the CPU baseline of rank 0
    pr = js["per_rank_solves_per_s_no_collective"]
    assert pr["batch_per_gpu"] == 8192 and 0 < pr["min"] <= pr["rank0"] <= pr["max"] or pr["min"] <= pr["max"]
    assert js["config"]["one_gpu_solves_per_s_at_this_batch"] == pr["rank0"] and js["config"]["batch_per_gpu"] == 8192
    assert "traffic" in js["roofline"] and "traffic_source" in js["roofline"]
    cb = js["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    # a world size that contradicts --gpus is refused by every rank
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), bench, "--gpus", "4", "--dry-collective", "gloo"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=280)
    assert r.returncode != 0 and b"n_gpus" not in r.stdout
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
        assert r.returncode == 3 and r.stdout.strip() == b"" and b"refusing" in r.stderr


def test_bench_traffic_bookkeeping():
    """bench.py's roofline.traffic plumbing (no GPU): the PMC child passes inherit the workload arguments and nothing else (own step
    counts, no side measurements, no nested passes), and a committed summary only stands in for the workload it was taken on -- the GP
    summary is tagged and never matches the plain N = 20 line (it once did: 73 MB reported for a 24 MB step)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    got = bench.traffic_child_args(["--steps", "20", "--warmup=5", "--horizon", "40", "--gp", "--no-cpu-baseline", "--batch-per-gpu", "8192"], 3, 1)
    assert got == ["--horizon", "40", "--gp", "--batch-per-gpu", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-two-in-flight", "--no-tight-stop", "--no-live-traffic"]
    plain, gp = bench.measured_traffic(20, 4096, "f64"), bench.measured_traffic(20, 4096, "f64", "gp")
    assert plain and gp and "gp" not in os.path.basename(plain["source"]) and os.path.basename(gp["source"]).startswith("gp")
    assert plain["bytes"] < 3 * 4096 * bench.algorithmic_bytes_per_solve(20) < gp["bytes"] * 3
    assert bench.measured_traffic(20, 4097, "f64") is None
