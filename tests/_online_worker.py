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
run = synth.make_online_run(n_nodes, n_pairs, n_kp=n_kp)
o = online.OnlineSlam(run, device=0, rank=rank, world=world, tdist=dist, match_batch=96, lm_iterations=6, match_cfg=dict(ransac_iteration=100))
o.upload_frames()
assert len(o.fid) < n_pairs          # this rank holds only its shard of the frames
o.run_all()
if rank == 0:
    np.savez(out, poses=o.poses, f_key=o.f_key, f_sticky=o.f_sticky, accept=np.array(o.accept_log), consensus=o.results["consensus"], T=o.results["T"])
dist.barrier()
print("ONLINE_OK world=%d rank=%d" % (world, rank))
o.close()
dist.destroy_process_group()
