"""Child process for the RCCL tests of tests/test_sharded_gpu.py (world_size 1 on cuda:0).  RCCL's bootstrap runs in a
process of its own so that a stalled bootstrap is a timed-out test, not a silent test run.
    _rccl_worker.py callback          torch.distributed "nccl" callback through every exchange step
    _rccl_worker.py native N E        communicator owned by the handle (uzl_rccl_unique_id + uzl_pgo_set_shard_rccl)"""
import os
import socket
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def callback():
    import torch
    import torch.distributed as dist
    from uzliti_slam_amd import capi, sharded, synth
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        g = synth.make_pose_graph(500, 2000, seed=5)
        poses, st = sharded.solve_sharded(capi, g, 0, 1, dist, torch, iterations=5, device=0, force_callback=True)
        ref = capi.Pgo()
        ref.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        st_ref = ref.optimize(5)
        pr, _, _ = ref.store()
        ref.close()
        # the exchange path keeps the level-0 smoother block-diagonal (ranks hold only their own edges' off-diagonal
        # blocks), so the iteration counts differ from the plain solve; the result does not
        assert st["status"] == 0 and st["iterations_done"] == st_ref["iterations_done"]
        dt, dr = synth.pose_errors(poses.reshape(-1, 3, 4), pr.reshape(-1, 3, 4))
        assert dt < 1e-4 and dr < 1e-5, (dt, dr)
    finally:
        dist.destroy_process_group()
    print("RCCL_CALLBACK_OK dt=%.2e dr=%.2e exchanges=%d" % (dt, dr, st["exchange_calls"]), flush=True)


def native(n, e):
    from uzliti_slam_amd import capi, synth
    from test_sharded_gpu import InProcessAllReduce
    g = synth.make_pose_graph(n, e, seed=5)
    ref = capi.Pgo()
    ref.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st_ref = ref.optimize(6)
    pr, _, _ = ref.store()
    ref.close()
    p = capi.Pgo()
    p.set_shard_rccl(0, 1, capi.rccl_unique_id())
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st = p.optimize(6)
    poses, _, _ = p.store()
    assert st["status"] == 0 and st["iterations_done"] == st_ref["iterations_done"]
    assert st["exchange_calls"] >= st["pcg_iterations"] > 0
    dt, dr = synth.pose_errors(poses.reshape(-1, 3, 4), pr.reshape(-1, 3, 4))
    assert dt < 1e-4 and dr < 1e-5, (dt, dr)
    # a second solve on the same communicator, then back to unsharded on the same handle
    p.reset()
    st2 = p.optimize(6)
    assert np.array_equal(p.store()[0], poses) and st2["pcg_iterations"] == st["pcg_iterations"]
    p.set_shard(0, 1, None)
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st3 = p.optimize(6)
    assert st3["exchange_calls"] == 0 and np.array_equal(p.store()[0], pr)
    p.close()
    # the same graph through the callback path: identical arithmetic, only the transport differs
    ar = InProcessAllReduce(1)
    q = capi.Pgo()
    q.set_shard(0, 1, ar.fn(0))
    q.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    q.optimize(6)
    assert np.array_equal(q.store()[0], poses)
    q.close()
    print("RCCL_NATIVE_OK n=%d e=%d dt=%.2e dr=%.2e exchanges=%d" % (n, e, dt, dr, st["exchange_calls"]), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "callback":
        callback()
    else:
        native(int(sys.argv[2]), int(sys.argv[3]))
