#!/usr/bin/env python3
"""Where the host side of a NEW structure spends its time (UZL_VERBOSE=1 prints build_structure's ticks): first solve of a fresh graph.
   UZL_VERBOSE=1 python tests/diag/structure_ticks.py nodes edges 2>&1 | grep structure"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth
n, e = int(sys.argv[1]), int(sys.argv[2])
w = capi.Pgo(iterations=1); gw = synth.make_pose_graph(200, 600, seed=1); w.add_graph(gw["nodes_pose"], gw["nodes_fixed"], gw["edges"]); w.optimize(1); w.close()
g = synth.make_pose_graph(n, e, seed=12345)
p = capi.Pgo(iterations=1)
t0 = time.perf_counter(); p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); t1 = time.perf_counter()
st = p.optimize(1); t2 = time.perf_counter()
sys.stderr.write("[diag] %d / %d: add_graph %.3f ms, first optimize(1) %.3f ms of which structure %.3f ms\n" % (n, e, 1e3 * (t1 - t0), 1e3 * (t2 - t1), st["structure_ms"]))
