"""Sharded single-graph pose-graph solve across the GPUs of one node (BASELINE config 4).

Every rank holds all vertices and linearises a contiguous range of the system edges; the exchange steps are
all-reduces of device buffers (RCCL over xGMI through torch.distributed, backend "nccl"):
  per linearisation : H_aa | b  (42 doubles per free vertex), chi2, the level-1 Galerkin arrays of the preconditioner
  per PCG iteration : [A p | restricted A p | p.Ap partials]  (about 6.8 doubles per free vertex: 0.5 MB at 10k vertices)
At pose-graph sizes these are latency-bound collectives (tens of microseconds each against ~15 us of compute per
iteration), so this mode is slower than one GPU (SURVEY §7 hard part 4): it exists for graphs that are too large
for the single-GPU path to be comfortable, not for speed.  Independent graphs / node-pair jobs scale through
uzliti_slam_amd/dist.py instead (no collective).
"""
import numpy as np


class _DevView:
    """Zero-copy view of `count` doubles at a raw device pointer (for torch.as_tensor)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(count),), "typestr": "<f8", "version": 2}


def make_rccl_allreduce(dist, torch, device=None):
    """all-reduce callback for capi.Pgo.set_shard built on torch.distributed (RCCL).  The solver's kernels run on
    the handle's own HIP stream, RCCL on torch's: both sides are fenced with device-wide synchronisation, which is
    correct but slow - the native exchange (`Pgo.set_shard_rccl`, the handle owns the communicator and issues
    ncclAllReduce on its own stream without any host synchronisation) is the product path; this callback remains for
    callers that already hold a torch process group.  `device` = the handle's HIP device ordinal (pinned here: a view made
    on another current device would silently become a copy and the reduced values would never reach the solver)."""
    dev = torch.cuda.current_device() if device is None else int(device)

    def allreduce(ptr, count, stream):
        torch.cuda.set_device(dev)
        torch.cuda.synchronize(dev)                    # the solver's stream has produced the buffer
        t = torch.as_tensor(_DevView(ptr, count), device="cuda:%d" % dev)
        if t.data_ptr() != int(ptr):
            return -1                                  # not a zero-copy view
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(dev)                    # reduced values visible before the solver's next kernel
        return 0
    return allreduce


def make_staged_allreduce(dist, torch):
    """Backend-agnostic variant: device -> host, all-reduce of a CPU tensor (gloo), host -> device.  Used by the
    two-process test on a one-GPU box (RCCL refuses two ranks on one device) and as a fallback without xGMI."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def allreduce(ptr, count, stream):
        host = np.empty(count, np.float64)
        if hip.hipStreamSynchronize(ctypes.c_void_p(stream)) != 0:
            return -1
        if hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(8 * count), 2) != 0:
            return -2
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if hip.hipMemcpy(ctypes.c_void_p(ptr), host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(8 * count), 1) != 0:
            return -3
        return 0
    return allreduce


def solve_sharded(capi, graph, rank, world, dist, torch, iterations=20, device=None, staged=False, force_callback=False):
    """Convenience: one rank's part of a sharded solve.  Returns (poses, stats) - identical on every rank."""
    dev = device if device is not None else rank
    p = capi.Pgo(device=dev)
    cb = None
    if world > 1 or force_callback:
        cb = make_staged_allreduce(dist, torch) if staged else make_rccl_allreduce(dist, torch, dev)
    p.set_shard(rank, world, cb)
    p.add_graph(graph["nodes_pose"], graph["nodes_fixed"], graph["edges"])
    st = p.optimize(iterations)
    poses, _, _ = p.store()
    p.close()
    return poses, st
