#!/usr/bin/env python3
"""Diagnostic: what an LM iteration costs apart from its PCG iterations.  Solves the same graph at the default accuracy and at an accuracy
so loose that every solve stops at its first look (8 iterations); the slope between the two is the cost of a PCG iteration, the intercept
what linearise / assemble / set-up / evaluate / host round trips cost per LM iteration.
  python tests/diag/lm_overhead.py N:E ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth   # noqa: E402

for a in sys.argv[1:] or ["1000:5000", "10000:50000"]:
    n, e = (int(x) for x in a.split(":"))
    g = synth.make_pose_graph(n, e)
    pts = []
    for tol in (1e-5, 1e-3):
        p = capi.Pgo(pcg_tol=tol)
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        p.optimize(20)
        best = 1e30
        for _ in range(5):
            p.reset(); t0 = time.perf_counter(); st = p.optimize(20); best = min(best, time.perf_counter() - t0)
        pts.append((st["pcg_iterations"], 1e3 * best, st["lm_trials"], st["precond_builds"]))
        p.close()
    (i0, t0_, tr0, b0), (i1, t1_, tr1, b1) = pts
    slope = (t0_ - t1_) / max(i0 - i1, 1)
    print("%d/%d: %.2f ms at %d PCG iterations (%d trials, %d rebuilds); %.2f ms at %d (%d trials, %d rebuilds) -> %.2f us per PCG iteration, %.0f us per LM trial besides"
          % (n, e, t0_, i0, tr0, b0, t1_, i1, tr1, b1, 1e3 * slope, 1e3 * (t1_ - slope * i1) / max(tr1, 1)))
