#!/usr/bin/env python3
"""Diagnostic: Q graphs of NODES / EDGES through R resident slots (uzl_pgo_batch_set_resident): aggregate edges/s.
   python tests/diag/batch_queue.py Q R [R ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

N = int(os.environ.get("NODES", "1000")); E = int(os.environ.get("EDGES", str(5 * N)))
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bt = capi.PgoBatch(Q)
for k in range(Q):
    g = synth.make_pose_graph(N, E, seed=12345 + 1000 * (k % 64) + k)
    bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
for R in [int(x) for x in sys.argv[2:]] or [16, 64]:
    bt.set_resident(R)
    bt.optimize(20)
    for p in bt.graphs:
        p.reset()
    t0 = time.perf_counter()
    sts = bt.optimize(20)
    dt = time.perf_counter() - t0
    edges = sum(st["n_edges"] * st["iterations_done"] for st in sts)
    tr = [st["lm_trials"] for st in sts]
    print("Q = %d, R = %3d: %.2f M edges/s, %.1f ms, batched %d, trials %d..%d, passes per graph %.1f" % (Q, R, edges / dt / 1e6, 1e3 * dt, bt.n_batched, min(tr), max(tr),
          sum(st["lm_passes"] for st in sts) / Q), flush=True)
bt.close()
