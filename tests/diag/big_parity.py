#!/usr/bin/env python3
"""Diagnostic: very large graphs (beyond what the CPU checker solves in reasonable time) - the multilevel path with its dense level-2
operator against the block-Jacobi path on the same problem, same LM iterations: pose difference, chi2, trial counts.
   python tests/diag/big_parity.py [n:e ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

shapes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(25000, 125000), (40000, 200000), (60000, 300000), (50000, 100000)]
bad = 0
for n, e in shapes:
    g = synth.make_pose_graph(n, e, seed=n % 97)
    res = {}
    for pre in (1, 0):
        p = capi.Pgo(preconditioner=pre)
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        t0 = time.perf_counter(); st = p.optimize(4); dt = time.perf_counter() - t0
        res[pre] = (p.store()[0].reshape(-1, 3, 4), st, dt)
        p.close()
    d_t, d_r = synth.pose_errors(res[1][0], res[0][0])
    a, b = res[1][1], res[0][1]
    ok = d_t < 1e-3 and d_r < 1e-4 and a["lm_trials"] == b["lm_trials"] and a["status"] == 0 and b["status"] == 0
    bad += 0 if ok else 1
    print("%s %6d/%7d: multilevel %7.1f ms (%5d pcg) | block-Jacobi %8.1f ms (%6d pcg) | dpose %.2e m %.2e rad  chi2 rel %.1e  trials %d/%d" % (
        "ok  " if ok else "MISS", n, e, 1e3 * res[1][2], a["pcg_iterations"], 1e3 * res[0][2], b["pcg_iterations"], d_t, d_r,
        abs(a["chi2_final"] - b["chi2_final"]) / b["chi2_final"], a["lm_trials"], b["lm_trials"]), flush=True)
print("%d shapes, %d misses" % (len(shapes), bad))
sys.exit(1 if bad else 0)
