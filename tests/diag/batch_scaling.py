#!/usr/bin/env python3
"""Diagnostic: aggregate edges/s of uzl_pgo_batch_* over B graphs of NODES / EDGES (environment; default config 2), B = 1, 2, 4, ... from argv."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

N = int(os.environ.get("NODES", "1000")); E = int(os.environ.get("EDGES", str(5 * N)))
for B in [int(x) for x in sys.argv[1:]] or [1, 4, 16, 64]:
    bt = capi.PgoBatch(B)
    for k in range(B):
        g = synth.make_pose_graph(N, E, seed=12345 + 1000 * k)
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    bt.optimize(20)
    reps = 5
    t0 = time.perf_counter(); edges = 0
    for _ in range(reps):
        for p in bt.graphs:
            p.reset()
        for st in bt.optimize(20):
            edges += st["n_edges"] * st["iterations_done"]
    dt = time.perf_counter() - t0
    print("B = %3d: %.2f M edges/s aggregate, %.2f ms per batch, %.3f ms per graph, batched %d" % (B, edges / dt / 1e6, 1e3 * dt / reps, 1e3 * dt / reps / B, bt.n_batched), flush=True)
    bt.close()
