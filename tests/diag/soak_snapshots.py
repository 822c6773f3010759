#!/usr/bin/env python3
"""Soak test of the pass driver's host / device hand-over (lm_tail_kernel's snapshot in pinned memory, read by the host once per pass):
thousands of short solves, each of which must reproduce the first one's poses and counters bit for bit - a snapshot read torn or early
would show as a different trial count, pass structure or result.
   python tests/diag/soak_snapshots.py [n_single=3000] [n_batches=150]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

n_single = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 150
KEYS = ("iterations_done", "lm_trials", "pcg_iterations", "terminated_early", "chi2_final", "lambda_final")
bad = 0
t0 = time.time()
for n, e, its in ((100, 300, 20), (400, 1800, 8), (900, 944, 10)):
    g = synth.make_pose_graph(n, e, seed=n + e)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st0 = p.optimize(its); ref = p.store()[0].copy()
    for k in range(n_single // 3):
        p.reset()
        st = p.optimize(its)
        if any(st[f] != st0[f] for f in KEYS) or not np.array_equal(p.store()[0], ref):
            bad += 1
            print("MISMATCH single %d/%d solve %d: %s vs %s" % (n, e, k, {f: st[f] for f in KEYS}, {f: st0[f] for f in KEYS}), flush=True)
    p.close()
    print("%d/%d: %d solves, %d mismatches so far (%.0f s)" % (n, e, n_single // 3, bad, time.time() - t0), flush=True)
B = 24
graphs = [synth.make_pose_graph(100 + 10 * (k % 5), 300 + 40 * (k % 7), seed=900 + k) for k in range(B)]
bt = capi.PgoBatch(B)
bt.set_resident(8)
for k, g in enumerate(graphs):
    bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
st0 = bt.optimize(12)
ref = [bt.graphs[k].store()[0].copy() for k in range(B)]
for r in range(n_batches):
    for k in range(B):
        bt.graphs[k].reset()
    st = bt.optimize(12)
    for k in range(B):
        if any(st[k][f] != st0[k][f] for f in KEYS) or not np.array_equal(bt.graphs[k].store()[0], ref[k]):
            bad += 1
            print("MISMATCH batch round %d graph %d" % (r, k), flush=True)
bt.close()
print("%d queued batches of %d graphs through 8 slots; total mismatches %d (%.0f s)" % (n_batches, B, bad, time.time() - t0))
sys.exit(1 if bad else 0)
