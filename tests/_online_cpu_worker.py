"""torchrun worker of tests/test_online_cpu.py: the online driver's multi-rank schedule (pair jobs sharded per batch, results gathered in
job order, gate / filter / solver on rank 0) on the CPU checker's stand-ins, gloo, no GPU."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import oracle as O                                  # noqa: E402
from online_stubs import oracle_online              # noqa: E402
from uzliti_slam_amd import synth                   # noqa: E402

out, n_nodes, n_pairs, n_kp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
run = synth.make_online_run(n_nodes, n_pairs, n_kp=n_kp)
o = oracle_online(O, run, ransac_iteration=60, rank=rank, world=world, tdist=dist, match_batch=40, lm_iterations=4, reopt_edges=64)
o.upload_frames()
assert 0 < len(o.fid) < n_pairs                     # this rank holds only its shard of the frames
o.run_all()
if rank == 0:
    np.savez(out, poses=o.poses, f_key=o.f_key, f_sticky=o.f_sticky, accept=np.array(o.accept_log), consensus=o.results["consensus"], T=o.results["T"],
             n_solves=len(o.solves))
dist.barrier()
print("ONLINE_CPU_OK world=%d rank=%d" % (world, rank))
dist.destroy_process_group()
