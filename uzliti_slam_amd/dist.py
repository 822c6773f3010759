"""Multi-GPU host logic: one process per GPU, independent units sharded across ranks.

Node-pair jobs and independent graphs are self-contained units, so the data path needs no collective: every rank
runs its shard through its own `capi.Match` / `capi.Pgo` handle (device = LOCAL_RANK) and only results / timings
travel.  `torch.distributed` (backend "nccl" = RCCL on the GPU node, "gloo" in the CPU tests) is plumbing for the
barrier, the max-over-ranks of a timed region and the result gather.
"""
import os

import numpy as np


def env_rank_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when run stand-alone."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_units, rank, world):
    """Contiguous, balanced partition of n_units: ranks [0, n % world) get one extra unit.  Returns (begin, end)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n_units), world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_pairs(pairs, rank, world):
    """Node-pair jobs of this rank, with their global job ids (the job id keys the RANSAC sampling stream, so the
    result of a pair does not depend on which GPU runs it or on how many GPUs there are)."""
    b, e = shard_range(len(pairs), rank, world)
    return list(pairs[b:e]), list(range(b, e))


def gather_edge_results(local_results, n_total, rank, world, dist=None):
    """All ranks' uzl_edge_result arrays, reassembled in global job order on every rank.
    local_results: structured numpy array (capi.EDGE_RESULT_DTYPE) of this rank's shard."""
    if world == 1 or dist is None:
        return local_results
    chunks = [None] * world
    dist.all_gather_object(chunks, np.ascontiguousarray(local_results).tobytes())
    out = np.concatenate([np.frombuffer(c, dtype=local_results.dtype) for c in chunks])
    assert len(out) == n_total, (len(out), n_total)
    for r in range(world):
        b, e = shard_range(n_total, r, world)
        assert e - b == len(chunks[r]) // local_results.dtype.itemsize
    return out


def max_over_ranks(value, dist=None, device=None):
    """max of a python float over ranks (the timed region of bench.py)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def replica_seed(base_seed, rank):
    """Independent graph per rank (disjoint subgraphs / local scopes): distinct generator seeds."""
    return int(base_seed) + int(rank)
