#!/usr/bin/env python3
"""Diagnostic: a batch of B config-2 graphs, timed, with the pass driver's segment times (UZL_PHASES=1, diagnostic build)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gs = [synth.make_pose_graph(1000, 5000, seed=12345 + 1000 * k) for k in range(B)]
bt = capi.PgoBatch(B)
for k, g in enumerate(gs):
    bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
bt.optimize(20)
ts = []
for _ in range(5):
    for g in bt.graphs:
        g.reset()
    t0 = time.perf_counter(); sts = bt.optimize(20); ts.append(time.perf_counter() - t0)
print("batch of %d: best %.2f ms median %.2f ms -> %.1f M edges/s; pcg per graph %s" % (B, 1e3 * min(ts), 1e3 * sorted(ts)[2], B * 5000 * 20 / sorted(ts)[2] / 1e6, [s["pcg_iterations"] for s in sts][:4]))
bt.close()
