"""torchrun worker of tests/test_sharded_gpu.py::test_native_rccl_world2: one process per GPU (LOCAL_RANK = device), the handle-owned RCCL
communicator (uzl_rccl_unique_id on rank 0, the bytes to the others through the gloo store, uzl_pgo_set_shard_rccl: ncclCommInitRank in
the library, ncclAllReduce on the solver's stream between its kernels).  Every rank writes its result; the parent compares.
    _rccl_world2_worker.py OUT_PREFIX N E ITERATIONS"""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uzliti_slam_amd import capi, synth   # noqa: E402

out, n, e, its = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dist.init_process_group("gloo")                       # rendezvous + the id broadcast only: the data path is the handle's own RCCL
rank, world = dist.get_rank(), dist.get_world_size()
dev = int(os.environ.get("LOCAL_RANK", rank))
assert capi.device_count() >= world, "one GPU per rank"
ids = [capi.rccl_unique_id() if rank == 0 else None]
dist.broadcast_object_list(ids, src=0)
g = synth.make_pose_graph(n, e, seed=5)
p = capi.Pgo(device=dev)
p.set_shard_rccl(rank, world, ids[0])
p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
st = p.optimize(its)
poses, err, _ = p.store()
assert st["status"] == 0 and st["exchange_calls"] >= st["pcg_iterations"] > 0
np.savez(out + ".rank%d.npz" % rank, poses=poses, err=err, pcg=st["pcg_iterations"], trials=st["lm_trials"], exchanges=st["exchange_calls"],
         n_eliminated=st["n_eliminated"], chi2=st["chi2_final"])
p.close()
dist.barrier()
print("RCCL_WORLD_OK world=%d rank=%d device=%d exchanges=%d" % (world, rank, dev, st["exchange_calls"]), flush=True)
dist.destroy_process_group()
