"""torchrun worker of tests/test_online_gpu.py::test_two_ranks_equal_one_rank: every rank matches its shard of each batch, rank 0
gates / filters / solves; the run's outcome (accepted edges, poses) is written by rank 0 as an .npz for the parent test."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uzliti_slam_amd import online, synth   # noqa: E402

out, n_nodes, n_pairs, n_kp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# UZL_ONE_GPU_PER_RANK=1 (test_online_two_ranks_native, boxes with >= 2 GPUs): rank r on device LOCAL_RANK; default: every rank on cuda:0
dev = int(os.environ.get("LOCAL_RANK", rank)) if os.environ.get("UZL_ONE_GPU_PER_RANK") else 0
run = synth.make_online_run(n_nodes, n_pairs, n_kp=n_kp)
o = online.OnlineSlam(run, device=dev, rank=rank, world=world, tdist=dist, match_batch=96, lm_iterations=6, match_cfg=dict(ransac_iteration=100))
o.upload_frames()
assert len(o.fid) < n_pairs          # this rank holds only its shard of the frames
o.run_all()
if rank == 0:
    np.savez(out, poses=o.poses, f_key=o.f_key, f_sticky=o.f_sticky, accept=np.array(o.accept_log), consensus=o.results["consensus"], T=o.results["T"])
dist.barrier()
print("ONLINE_OK world=%d rank=%d device=%d" % (world, rank, dev))
o.close()
dist.destroy_process_group()
