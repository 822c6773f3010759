#!/usr/bin/env python3
"""Diagnostic: the device-resident LM loop pass by pass (UZL_VERBOSE log of the last of a few repeated solves), and its wall time next to
the host-driven loop's.   python tests/diag/lm_passes.py [N:E ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth   # noqa: E402

for a in sys.argv[1:] or ["1000:5000"]:
    n, e = (int(x) for x in a.split(":"))
    g = synth.make_pose_graph(n, e)
    for loop in [int(x) for x in os.environ.get("LOOPS", "0,1").split(",")]:
        p = capi.Pgo(lm_loop=loop)
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        p.optimize(20)
        ts = []
        for _ in range(int(os.environ.get("REPS", "10"))):
            p.reset(); t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
        print("%d/%d lm_loop=%d: best %.3f ms  median %.3f ms  passes %d trials %d pcg %d" % (n, e, loop, 1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2], st["lm_passes"], st["lm_trials"], st["pcg_iterations"]), flush=True)
        if loop == 0 and os.environ.get("LM_LOG"):
            p.set_config(verbose=1)
            p.reset(); p.optimize(20)
        p.close()
