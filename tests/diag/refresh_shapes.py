#!/usr/bin/env python3
"""Diagnostic (diagnostic build, UZL_ML_REFRESH_REL in the environment): solve time and PCG iterations of a list of graph shapes - the
lazy-refresh threshold's effect beyond the two benchmark graphs.   UZL_ML_REFRESH_REL=3e-2 python tests/diag/refresh_shapes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

shapes = [(3000, 15000, 1), (5000, 25000, 2), (6000, 30000, 5), (10000, 50000, 12345), (10000, 50000, 7), (14000, 60000, 6), (20000, 100000, 3),
          (8000, 24000, 4), (20000, 21700, 3), (8000, 8400, 9), (12000, 12700, 2), (8000, 9000, 5), (20000, 24000, 6), (30000, 150000, 8)]
if len(sys.argv) > 1 and sys.argv[1] == "small":      # the small-graph class (rebuilds run ahead on the second stream): config 2 and its neighbours
    shapes = [(1000, 5000, 12345), (1000, 5000, 3), (600, 2600, 300), (2000, 9000, 11), (2500, 12000, 4), (300, 1200, 1500), (100, 300, 400),
              (1500, 1530, 3), (3000, 3100, 3), (600, 630, 3)]
tot = 0.
for n, e, seed in shapes:
    g = synth.make_pose_graph(n, e, seed=seed)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    p.optimize(20)
    ts = []
    for _ in range(9 if n <= 3000 else 3):
        p.reset()
        t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
    ms = 1e3 * float(np.median(ts)); tot += ms
    print("%6d/%6d seed %5d: %8.2f ms  pcg %5d  trials %d  builds %d  chi2 %.6g" % (n, e, seed, ms, st["pcg_iterations"], st["lm_trials"], st["precond_builds"], st["chi2_final"]), flush=True)
    p.close()
print("sum %.1f ms" % tot)
