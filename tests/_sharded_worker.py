"""torchrun worker for tests/test_sharded_gpu.py: world_size ranks, all on cuda:0, gloo for the exchange."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from uzliti_slam_amd import capi, sharded, synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, e, its = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    g = synth.make_pose_graph(n, e, seed=99)
    poses, st = sharded.solve_sharded(capi, g, rank, world, dist, torch, iterations=its, device=0, staged=True)
    # all ranks hold the same answer
    t = torch.from_numpy(poses.copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi), "ranks disagree"
    if rank == 0:
        ref = capi.Pgo(device=0)
        ref.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        st_ref = ref.optimize(its)
        pr, _, _ = ref.store()
        ref.close()
        dt, dr = synth.pose_errors(poses.reshape(-1, 3, 4), pr.reshape(-1, 3, 4))
        assert st["status"] == 0 and st["iterations_done"] == st_ref["iterations_done"], (st, st_ref)
        assert dt < 1e-5 and dr < 1e-6, (dt, dr)
        print("SHARDED_OK world=%d dt=%.2e dr=%.2e pcg=%d" % (world, dt, dr, st["pcg_iterations"]), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
