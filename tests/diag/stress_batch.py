#!/usr/bin/env python3
"""Diagnostic: randomized batches against single solves (bit-identical poses and iteration counts required)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth              # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = tot = nb_tot = 0
for b in range(n_batches):
    n = int(rng.choice([120, 400, 900, 1500, 2040]))
    B = int(rng.integers(2, 9)) if rng.random() < 0.5 else int(rng.integers(12, 25))     # (from 12 graphs on: two launch sequences)
    its = int(rng.choice([3, 8, 20]))
    graphs = []
    one_class = rng.random() < 0.7                     # most batches: one density class, so that the graphs share a hierarchy shape and batch
    dens0 = float(rng.choice([1.02, 1.5, 3.0, 5.0]))
    for k in range(B):
        dens = dens0 * (1. + 0.01 * float(rng.integers(0, 4))) if one_class else float(rng.choice([1.0, 1.02, 1.5, 3.0, 5.0]))
        g = synth.make_pose_graph(n, max(n - 1, int(n * dens)), seed=int(rng.integers(1, 10**6)), outlier_frac=float(rng.choice([0.0, 0.05, 0.3])))
        if rng.random() < 0.2:
            g = synth.permute_graph(g, rng.permutation(n))
        graphs.append(g)
    bt = capi.PgoBatch(B)
    if rng.random() < 0.3:
        bt.set_resident(int(rng.integers(1, B + 1)))
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st = bt.optimize(its)
    nb_tot += bt.n_batched
    for k, g in enumerate(graphs):
        p = capi.Pgo(); p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); s1 = p.optimize(its); P1 = p.store()[0]; p.close()
        ok = np.array_equal(bt.graphs[k].store()[0], P1) and all(st[k][f] == s1[f] for f in ("iterations_done", "lm_trials", "pcg_iterations", "chi2_final", "terminated_early"))
        tot += 1; bad += 0 if ok else 1
        if not ok:
            print("MISS batch %d graph %d n %d its %d: batch (%d its, %d pcg, chi2 %.9g) single (%d, %d, %.9g)" % (b, k, n, its, st[k]["iterations_done"], st[k]["pcg_iterations"],
                  st[k]["chi2_final"], s1["iterations_done"], s1["pcg_iterations"], s1["chi2_final"]), flush=True)
    bt.close()
print("%d graphs in %d batches (%d solved batched), %d misses" % (tot, n_batches, nb_tot, bad))
sys.exit(1 if bad else 0)
