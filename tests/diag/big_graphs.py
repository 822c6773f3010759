#!/usr/bin/env python3
"""Diagnostic: solve time of very large loopy graphs (beyond BASELINE's sizes) - where the dense level-2 operator (UZL_ML_COMP4_MAX in the
diagnostic build) stops paying.   python tests/diag/big_graphs.py [n:e ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

shapes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(30000, 150000), (40000, 200000), (60000, 300000)]
for n, e in shapes:
    t0 = time.time(); g = synth.make_pose_graph(n, e, seed=8); tg = time.time() - t0
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    t0 = time.perf_counter(); st = p.optimize(20); t1 = time.perf_counter()
    p.reset()
    t2 = time.perf_counter(); st2 = p.optimize(20); t3 = time.perf_counter()
    print("%6d/%7d: first %8.1f ms, again %8.1f ms  pcg %6d  trials %d  builds %d  status %d  chi2 %.6g   (synth %.1f s)" % (n, e, 1e3 * (t1 - t0), 1e3 * (t3 - t2), st2["pcg_iterations"], st2["lm_trials"], st2["precond_builds"], st2["status"], st2["chi2_final"], tg), flush=True)
    p.close()
