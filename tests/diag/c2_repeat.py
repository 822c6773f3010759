#!/usr/bin/env python3
"""Diagnostic: config-2 / config-4 solve time, repeated (best / median of n), to separate code changes from box-to-box noise."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

for n, e, reps in ((1000, 5000, 40), (10000, 50000, 6)):
    g = synth.make_pose_graph(n, e)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    p.optimize(20)
    ts = []
    for _ in range(reps):
        p.reset()
        t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("%d/%d: best %.3f ms median %.3f ms  pcg %d  -> %.2f M edges/s (median)" % (n, e, ts.min(), np.median(ts), st["pcg_iterations"], st["n_edges"] * 20 / np.median(ts) / 1e3), flush=True)
    p.close()
