#!/usr/bin/env python3
"""Diagnostic: small chain-like graphs with the reduced system in row order / by strong aggregates (UZL_SCHUR_STRONG_MIN lowered, diagnostic build)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

for n, e in ((600, 630), (1500, 1530), (1500, 1600), (2500, 2600)):
    g = synth.make_pose_graph(n, e, seed=n)
    for mode in (1, 2):
        p = capi.Pgo(reduced_numbering=mode)
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        p.optimize(20)
        ts = []
        for _ in range(20):
            p.reset()
            t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
        print("%d/%d numbering %d: median %.3f ms  pcg %d  eliminated %d strong %d" % (n, e, mode, 1e3 * np.median(ts), st["pcg_iterations"], st["n_eliminated"], st["reduced_strong"]), flush=True)
        p.close()
