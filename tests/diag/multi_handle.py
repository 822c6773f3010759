#!/usr/bin/env python3
"""Diagnostic: B solver handles, one host thread each, sharing one GPU (the "one graph per handle" way of filling the device).
  python tests/diag/multi_handle.py [B ...]        aggregate edges/s over B concurrent config-2 solves"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

Bs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16]
reps = 5
for B in Bs:
    hs = []
    for b in range(B):
        g = synth.make_pose_graph(1000, 5000, seed=12345 + b)
        p = capi.Pgo()
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        p.optimize(20)
        hs.append(p)
    bar = threading.Barrier(B + 1)
    edges = [0] * B

    def work(i):
        bar.wait()
        for _ in range(reps):
            hs[i].reset()
            st = hs[i].optimize(20)
            edges[i] += st["n_edges"] * st["iterations_done"]
        bar.wait()

    th = [threading.Thread(target=work, args=(i,)) for i in range(B)]
    for t in th:
        t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in th:
        t.join()
    print("B = %2d handles: %.2f M edges/s aggregate, %.2f ms per solve per handle" % (B, sum(edges) / dt / 1e6, 1e3 * dt / reps), flush=True)
    for p in hs:
        p.close()
