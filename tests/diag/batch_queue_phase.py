#!/usr/bin/env python3
"""Diagnostic: batches created after k other single-stream handles (k = 0..7), i.e. at every phase of the runtime's stream -> hardware queue
assignment: the streams of a batch must not share a queue (uzl_pgo.hip: overlapping_stream), so the rate must not depend on k.
  python tests/diag/batch_queue_phase.py [chain|c2|small]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "c2"
if kind == "chain":
    gs = [synth.make_pose_graph(1500, 1530, seed=4040 + k) for k in range(16)]
elif kind == "small":
    gs = [synth.make_pose_graph(100, 300, seed=777 + 1000 * k) for k in range(64)]
else:
    gs = [synth.make_pose_graph(1000, 5000, seed=12345 + 1000 * k) for k in range(16)]
B = len(gs)
for k in range(8):
    others = [capi.Match(device=0, ransac_threshold=0.1, ransac_iteration=100, ransac_break_percentage=0.6, seed=1) for _ in range(k)]     # one stream each
    bt = capi.PgoBatch(B)
    for i, g in enumerate(gs):
        bt.graphs[i].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    bt.optimize(20)
    ts = []
    for _ in range(3):
        for g in bt.graphs: g.reset()
        t0 = time.perf_counter(); st = bt.optimize(20); ts.append(time.perf_counter() - t0)
    e = sum(x["n_edges"] * x["iterations_done"] for x in st)
    print("%s, %d other stream(s) first: batch of %d median %.2f ms -> %.1f M edges/s" % (kind, k, B, 1e3 * sorted(ts)[1], e / sorted(ts)[1] / 1e6), flush=True)
    bt.close()
    for m in others: m.close()
