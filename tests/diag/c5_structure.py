#!/usr/bin/env python3
"""Diagnostic: where add_graph + structure of a config-5 re-optimisation go (UZL_VERBOSE ticks of build_structure) - the last interval's
input (gpurun_out/c5_last.npz written by tests/diag/c5_last.py, or a synthetic chain-like 20000 / 21800 graph)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth   # noqa: E402

g = synth.make_pose_graph(20000, 21800, seed=3)
p = capi.Pgo()
for rep in range(3):
    t0 = time.perf_counter()
    # a changed structure every time: drop a different loop closure
    e = {k: v.copy() for k, v in g["edges"].items()}
    e["valid"][20000 + rep] = 0
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], e)
    t1 = time.perf_counter()
    if rep == 2:
        p.set_config(verbose=1)
    st = p.optimize(1)
    t2 = time.perf_counter()
    print("rep %d: add_graph %.2f ms, optimize(1) %.2f ms of which structure %.2f ms (reused %d)" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), st["structure_ms"], st["structure_reused"]), flush=True)
p.close()
