#!/usr/bin/env python3
"""Diagnostic: per-graph statistics of the 16 config-2 graphs of the batched bench block, alone and in the batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gs = [synth.make_pose_graph(1000, 5000, seed=12345 + 1000 * k) for k in range(B)]
single = []
for g in gs:
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    single.append(p.optimize(20))
    p.close()
bt = capi.PgoBatch(B)
for k, g in enumerate(gs):
    bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
sts = bt.optimize(20)
for k in range(B):
    a, b = single[k], sts[k]
    print("graph %2d: alone pcg %4d trials %2d builds %d status %d | batch pcg %4d trials %2d status %d" % (
        k, a["pcg_iterations"], a["lm_trials"], a.get("preconditioner_builds", -1), a["status"], b["pcg_iterations"], b["lm_trials"], b["status"]))
print("sum alone %d, max-per-round proxy: max %d" % (sum(a["pcg_iterations"] for a in single), max(a["pcg_iterations"] for a in single)))
bt.close()
