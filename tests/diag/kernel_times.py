#!/usr/bin/env python3
"""Diagnostic: per-kernel times of one solve (HIP-event timing inside the library) for a list of graph shapes.
  python tests/diag/kernel_times.py 10000:50000 20000:100000 20000:21700 [its=20]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth              # noqa: E402

its = 20
shapes = []
for a in sys.argv[1:]:
    if a.startswith("its="):
        its = int(a[4:])
    else:
        n, e = a.split(":"); shapes.append((int(n), int(e)))
for n, e in shapes:
    g = synth.make_pose_graph(n, e)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    p.optimize(its)                                   # warm-up
    p.reset()
    t0 = time.time(); st = p.optimize(its); wall = 1e3 * (time.time() - t0)
    p.reset()
    p.set_profiling(True)
    st2 = p.optimize(its)
    kt = p.kernel_times()
    p.close()
    print("n %d e %d: %.1f ms  pcg %d  trials %d  status %d" % (n, e, wall, st["pcg_iterations"], st["lm_trials"], st["status"]))
    for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])[:8]:
        print("   %-14s %9.3f ms %7d launches %8.2f us" % (k, v["ms"], v["launches"], 1e3 * v["ms"] / max(v["launches"], 1)))
