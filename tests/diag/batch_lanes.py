#!/usr/bin/env python3
"""Diagnostic: B config-2 graphs as ONE batch against the same graphs as L batches of B / L driven from L host threads (each batch has its
own streams and slot table): do the lanes' launch sequences overlap on the GPU?   python tests/diag/batch_lanes.py [B] [L ...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lanes = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
gs = [synth.make_pose_graph(1000, 5000, seed=12345 + 1000 * k) for k in range(B)]
for L in lanes:
    per = B // L
    bts = [capi.PgoBatch(per) for _ in range(L)]
    for k, g in enumerate(gs[:per * L]):
        bts[k // per].graphs[k % per].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])

    def run(bt):
        bt.optimize(20)

    def all_lanes():
        th = [threading.Thread(target=run, args=(bt,)) for bt in bts[1:]]
        for t in th: t.start()
        run(bts[0])
        for t in th: t.join()

    all_lanes()
    ts = []
    for _ in range(5):
        for bt in bts:
            for g in bt.graphs: g.reset()
        t0 = time.perf_counter(); all_lanes(); ts.append(time.perf_counter() - t0)
    med = sorted(ts)[2]
    print("%d graphs as %d lane(s) of %d: best %.2f ms median %.2f ms -> %.1f M edges/s" % (per * L, L, per, 1e3 * min(ts), 1e3 * med, per * L * 5000 * 20 / med / 1e6), flush=True)
    for bt in bts: bt.close()
