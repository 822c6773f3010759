"""GPU test of the sharded single-graph solve (BASELINE config 4's exchange pattern) on ONE device: two handles act
as rank 0 / rank 1, each linearising half of the edges, with an in-process all-reduce (device -> host, barrier, sum,
host -> device) standing in for RCCL.  The result must equal the unsharded solve and the oracle."""
import ctypes
import threading

import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu


class InProcessAllReduce:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.bufs = [None] * world
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.calls = 0
        self.doubles = 0

    def fn(self, rank):
        def allreduce(ptr, count, stream):
            host = np.empty(count, np.float64)
            assert self.hip.hipStreamSynchronize(ctypes.c_void_p(stream)) == 0
            assert self.hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(8 * count), 2) == 0
            self.bufs[rank] = host
            self.barrier.wait(timeout=120)
            total = self.bufs[0].copy()
            for r in range(1, self.world):          # fixed order: every rank gets bit-identical sums
                total += self.bufs[r]
            self.barrier.wait(timeout=120)
            assert self.hip.hipMemcpy(ctypes.c_void_p(ptr), total.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(8 * count), 1) == 0
            if rank == 0:
                self.calls += 1; self.doubles += count
            return 0
        return allreduce


@pytest.mark.parametrize("n,e,world", [(300, 1200, 2), (1000, 5000, 2), (600, 2500, 3), (3000, 12000, 2), (10000, 50000, 2),   # BASELINE config 4 at its size
                                       (13000, 16000, 2),        # the ml_alpha_kernel path (12k .. 21.8k vertices)
                                       (2000, 2040, 2), (20000, 21800, 2), (5000, 5400, 3)])      # chain-like: chain interiors Schur-eliminated by every rank
def test_sharded_equals_unsharded(capi, oracle, n, e, world):
    g = synth.make_pose_graph(n, e, seed=n + world)
    ref = capi.Pgo()
    ref.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st_ref = ref.optimize(8)
    poses_ref, _, _ = ref.store()
    ref.close()

    ar = InProcessAllReduce(world)
    out = [None] * world

    def run(rank):
        try:
            p = capi.Pgo()
            p.set_shard(rank, world, ar.fn(rank))
            p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
            st = p.optimize(8)
            poses, err, used = p.store()
            p.close()
            out[rank] = (poses, st, err)
        except Exception as ex:      # pragma: no cover
            out[rank] = ex
            ar.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for o in out:
        assert not isinstance(o, Exception), o
    for r in range(world):
        poses, st, err = out[r]
        assert st["status"] == 0 and st["iterations_done"] == st_ref["iterations_done"]
        assert st["n_eliminated"] == st_ref["n_eliminated"] and (st["n_eliminated"] > 0) == (e < 2 * n)      # the sharded solve reduces what the plain one does
        assert abs(st["chi2_initial"] - st_ref["chi2_initial"]) <= 1e-9 * st_ref["chi2_initial"]
        assert st["exchange_calls"] == ar.calls and st["exchange_ms"] > 0          # uzl_pgo_stats accounts for the exchange
        dt, dr = synth.pose_errors(poses.reshape(-1, 3, 4), poses_ref.reshape(-1, 3, 4))
        assert dt < 1e-4 and dr < 1e-5, (r, dt, dr)              # same LM, PCG stopped at the same tolerance
        assert np.array_equal(poses, out[0][0])                  # every rank ends with bit-identical poses
    # against the CPU checker at every size, BASELINE config 4's 10k/50k included (its direct solve: ~8 s per 8 iterations)
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, _ = oracle.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=8)
    dt, dr = synth.pose_errors(out[0][0].reshape(-1, 3, 4), P.reshape(-1, 3, 4))
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    # exchange volume: one all-reduce per PCG iteration (iterations are enqueued in batches of 16, so up to 15 run
    # past convergence per solve) + a handful per LM trial (H_aa|b, chi2, level-1 Galerkin arrays)
    pcg = out[0][1]["pcg_iterations"]
    assert pcg <= ar.calls <= pcg + 8 * 3 * (16 + 6), (pcg, ar.calls)


def test_sharded_two_processes(capi):
    """Real multi-process path: torchrun, 2 ranks (both on cuda:0 of the one-GPU box), gloo-staged all-reduce."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(here, "_sharded_worker.py"), "400", "1600", "5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "SHARDED_OK world=2" in r.stdout


def _rccl_child(*args):
    """RCCL's bootstrap in a child process with a deadline: a stall there fails this test instead of hanging the run."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")           # one node: the bootstrap needs no external interface
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_worker.py")] + [str(a) for a in args],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_rccl_callback_world1(capi):
    """The RCCL callback (torch.distributed backend "nccl", zero-copy view of the solver's device buffer) driven
    through every exchange step of a solve with world_size 1: the result must equal the plain solve."""
    assert "RCCL_CALLBACK_OK" in _rccl_child("callback")


@pytest.mark.parametrize("n,e", [(500, 2000), (3000, 12000)])
def test_native_rccl_world1(capi, n, e):
    """The exchange owned by the handle: uzl_rccl_unique_id + uzl_pgo_set_shard_rccl (ncclCommInitRank inside the library,
    ncclAllReduce on the solver's own stream between its kernels, no callback, no host synchronisation), driven through every
    exchange step of a solve with world_size 1.  Must equal the plain solve; the same graph through the callback path must give
    the same bits (identical arithmetic, only the transport differs)."""
    assert "RCCL_NATIVE_OK" in _rccl_child("native", n, e)


def _torchrun(worker, nproc, *args, timeout=900):
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, worker)] + [str(a) for a in args],
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


@pytest.mark.parametrize("n,e", [(3000, 12000), (10000, 50000), (20000, 21800)])
def test_native_rccl_world2(capi, oracle, tmp_path, n, e):
    """SURVEY 8(e) row 3 on REAL links: two processes, two GPUs, the handle-owned RCCL communicator with world_size 2 - ncclAllReduce
    between ranks on every exchange step (per linearisation, per LM trial, per PCG iteration).  Skipped on a one-GPU box: the first
    multi-GPU box that runs the suite validates the path by itself.  Every rank must end with bit-identical poses (same numbers after
    every exchange, same scalar logic), within the north-star bar of the oracle's direct solve, and - for the chain-like graph - with its
    chain interiors Schur-eliminated."""
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU, RCCL between them)")
    its = 6
    out = str(tmp_path / "w2")
    stdout = _torchrun("_rccl_world2_worker.py", 2, out, n, e, its)
    assert stdout.count("RCCL_WORLD_OK world=2") == 2
    z0, z1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    assert np.array_equal(z0["poses"], z1["poses"]) and int(z0["pcg"]) == int(z1["pcg"]) and int(z0["trials"]) == int(z1["trials"])
    g = synth.make_pose_graph(n, e, seed=5)
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, _ = oracle.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=its)
    dt, dr = synth.pose_errors(z0["poses"].reshape(-1, 3, 4), P.reshape(-1, 3, 4))
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    if e < 2 * n:
        assert int(z0["n_eliminated"]) > 0

