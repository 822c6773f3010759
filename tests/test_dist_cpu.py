"""CPU tests of the multi-GPU host logic with world_size 2 over gloo: sharding of independent node-pair
jobs, result gather in global job order, max/sum over ranks.  The per-rank compute is stood in for by the CPU
oracle (tests may use it; the product path has no CPU compute)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from uzliti_slam_amd import dist as D
    for n in (0, 1, 7, 8, 512, 4097):
        for world in (1, 2, 3, 8):
            got = []
            sizes = []
            for r in range(world):
                b, e = D.shard_range(n, r, world)
                got += list(range(b, e)); sizes.append(e - b)
            assert got == list(range(n))
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import torch.distributed as dist
        import oracle as O
        from uzliti_slam_amd import capi, synth, dist as D
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert D.env_rank_world() == (rank, rank, world)
        pairs = synth.make_pairs(5, n_kp=120, seed=3)            # same list on every rank (same seed)
        mine, job_ids = D.shard_pairs(pairs, rank, world)
        res = np.zeros(len(mine), capi.EDGE_RESULT_DTYPE)
        for k, ((f, t, _), jid) in enumerate(zip(mine, job_ids)):
            w = O.estimate_edge([f], [t], ransac_threshold=0.1, ransac_iteration=60, break_percentage=0.6, seed=9, job_id=jid)
            res[k]["job_id"] = jid; res[k]["ok"] = w["ok"]; res[k]["consensus"] = w["consensus"]
            res[k]["T"] = w["T"].reshape(12); res[k]["mse"] = w["mse"]
        allres = D.gather_edge_results(res, len(pairs), rank, world, dist)
        t = D.max_over_ranks(1.0 + rank, dist)
        s = D.sum_over_ranks(float(len(mine)), dist)
        dist.barrier()
        q.put((rank, allres["job_id"].tolist(), allres["consensus"].tolist(), allres["T"].copy(), t, s))
        dist.destroy_process_group()
    except Exception as e:      # pragma: no cover
        q.put((rank, "error", repr(e)))


def test_world_size_2_gloo_job_sharding(oracle):
    import torch.multiprocessing as mp
    from uzliti_slam_amd import synth
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(o[1] != "error" for o in outs), outs
    pairs = synth.make_pairs(5, n_kp=120, seed=3)
    want = [oracle.estimate_edge([f], [t], ransac_threshold=0.1, ransac_iteration=60, break_percentage=0.6, seed=9, job_id=j)
            for j, (f, t, _) in enumerate(pairs)]
    for rank, job_ids, cons, T, tmax, ssum in outs:
        assert job_ids == [0, 1, 2, 3, 4]                        # global job order on every rank
        assert cons == [w["consensus"] for w in want]             # result independent of the rank that ran the pair
        assert np.array_equal(T, np.stack([w["T"].reshape(12) for w in want]))
        assert tmax == 2.0 and ssum == 5.0
