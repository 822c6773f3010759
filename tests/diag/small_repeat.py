#!/usr/bin/env python3
"""Diagnostic: solve time of small graphs (BASELINE config 1 and neighbours), repeated, with the pass driver's segment times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

for n, e in ((100, 300), (300, 1200), (600, 630)):
    g = synth.make_pose_graph(n, e)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    p.optimize(20)
    ts = []
    for _ in range(40):
        p.reset()
        t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("%d/%d: best %.3f ms median %.3f ms  pcg %d trials %d passes %d eliminated %d -> %.2f M edges/s (median)" % (n, e, ts.min(), np.median(ts), st["pcg_iterations"], st["lm_trials"], st["lm_passes"], st["n_eliminated"], st["n_edges"] * 20 / np.median(ts) / 1e3), flush=True)
    p.close()
