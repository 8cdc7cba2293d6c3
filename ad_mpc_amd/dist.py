"""Multi-GPU layer: scenario sharding and the one collective of the path (the final arg-min).

Instances are independent: rank r owns the contiguous global index range
[r*B/W, (r+1)*B/W) and solves it locally with no data-path collective.  The only exchange is the
16-byte (cost, global index) pair per rank, all-gathered over RCCL (backend "nccl" on ROCm) or gloo
in the CPU tests, followed by a local scan.  Tie-break: lowest global index; NaN never wins.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` instances for `rank` of `world` (remainder spread over the first ranks)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pick_min(vals, idxs):
    """arg-min over (value, index) pairs with the path's tie-break rules; returns 0-dim tensors."""
    vals = torch.where(torch.isnan(vals), torch.full_like(vals, float("inf")), vals)
    m = vals.min()
    cand = torch.where(vals == m, idxs, torch.full_like(idxs, torch.iinfo(torch.int64).max))
    return m, cand.min()


def local_argmin_torch(cost, index_offset=0):
    """Host-side (torch) equivalent of admpc_argmin used by the gloo CPU tests of the N>1 path."""
    idxs = torch.arange(cost.shape[0], dtype=torch.int64, device=cost.device) + int(index_offset)
    v, i = pick_min(cost, idxs)
    return v.reshape(1), i.reshape(1)


def global_argmin(val, idx, group=None):
    """Combine per-rank (val[1] f64, idx[1] i64) pairs: all-gather 16 B per rank, local scan.
    Every rank returns the same (value, global index)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return val.reshape(()), idx.reshape(())
    world = dist.get_world_size(group)
    mine = torch.cat([val.reshape(1).to(torch.float64), idx.reshape(1).to(torch.int64).view(torch.float64)])
    out = torch.empty(2 * world, dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    pairs = out.view(world, 2)
    vals = pairs[:, 0].contiguous()
    idxs = pairs[:, 1].contiguous().view(torch.int64)
    return pick_min(vals, idxs)


def pack_pair(val, idx):
    """The 16-byte record of the exchange: float64[2] = (value, bit pattern of the int64 global index)."""
    return torch.cat([val.reshape(1).to(torch.float64), idx.reshape(1).to(torch.int64).view(torch.float64)])


def pairs_min_torch(pairs):
    """torch restatement of admpc_argmin_pairs (NaN -> +inf, ties -> lowest global index) on float64[W,2] records.  Kept as an
    independent second statement of the rules: tests/argmin_spec.py checks it, the library's host reducer and the device
    kernels against one table."""
    vals = pairs[:, 0].contiguous()
    idxs = pairs[:, 1].contiguous().view(torch.int64)
    v, i = pick_min(vals, idxs)
    i = torch.where(i == torch.iinfo(torch.int64).max, torch.zeros_like(i), i)       # nothing but NaN / +inf records without an index
    return pack_pair(v, i)


def pairs_min_host(pairs):
    """Second-level reducer for records gathered into HOST memory (gloo, MPI): libadmpc's admpc_argmin_pairs_host, i.e. the same
    source (csrc/argmin_rule.h) the device kernel admpc_argmin_pairs_kernel is compiled from.  float64[W,2] CPU tensor in,
    float64[2] record out."""
    import ctypes as C
    from . import _lib
    if pairs.device.type != "cpu" or pairs.dtype != torch.float64 or pairs.dim() != 2 or pairs.shape[1] != 2:
        raise ValueError("pairs_min_host: float64[W,2] CPU tensor expected")
    pairs = pairs.contiguous()
    out = torch.empty(2, dtype=torch.float64)
    _lib.check(_lib.load().admpc_argmin_pairs_host(C.c_void_p(pairs.data_ptr()), int(pairs.shape[0]),
                                                   C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 8)))
    return out


def global_argmin_records(pair, reduce_pairs, group=None, gathered=None):
    """The record path of the N-GPU arg-min, whatever the backend: this rank's 16-byte record, all-gather of the records (one
    collective, 16 B per rank), second-level arg-min `reduce_pairs` over the gathered float64[W,2] records.  Every rank returns
    the same float64[2] record."""
    if not (dist.is_available() and dist.is_initialized()):
        return pair
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world, 2), dtype=torch.float64, device=pair.device)
    dist.all_gather_into_tensor(gathered.view(-1), pair, group=group)
    return reduce_pairs(gathered)


def global_argmin_device(engine, cost, index_offset=0, group=None, gathered=None):
    """The N-GPU arg-min in three device operations and no host synchronisation: admpc_argmin into a 16-byte record,
    all-gather of the records (RCCL), admpc_argmin_pairs.  Returns a float64[2] device tensor
    (value, bits of the int64 global index); `unpack_pair` reads it on the host."""
    return global_argmin_records(engine.argmin_pair(cost, index_offset), engine.argmin_pairs, group=group, gathered=gathered)


def unpack_pair(pair):
    """(value, global index) of a float64[2] record on the host (synchronises)."""
    h = pair.detach().cpu()
    return float(h[0]), int(h[1:2].view(torch.int64)[0])

