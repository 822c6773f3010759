#!/usr/bin/env python3
"""Diagnostic: randomized solver-vs-oracle sweep over graph sizes, loop densities, outlier fractions, orders and xy-only.
  python tests/diag/stress_pgo.py [n_cases] [seed] [large]     prints one line per case and a summary; exit code 1 on any miss
  `large`: sizes 7000 .. 24000 at 1.01 .. 1.3 edges per node (the oracle's direct solve stays fast on sparse graphs) - the large-graph
  kernel paths (rows of the level-2 operator in registers / ml_alpha_kernel / no dense level-2 operator)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O                                   # noqa: E402
from uzliti_slam_amd import capi, synth              # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"
ONLY = int(os.environ.get("ONLY", "-1"))          # run case ONLY alone (same random stream), with the solver's trial log
bad = 0
p = capi.Pgo(reduced_numbering=int(os.environ.get("NUMBERING", "0")))      # 0 = the handle chooses (with the history of the cases before), 1 / 2 = fixed
for k in range(n_cases):
    if LARGE:
        n = int(rng.choice([7000, 9500, 12000, 12500, 15000, 19000, 23000]))
        dens = float(rng.choice([1.01, 1.05, 1.15, 1.3]))
    else:
        n = int(rng.choice([150, 400, 900, 1500, 2300, 3500, 5000]))
        dens = float(rng.choice([1.01, 1.05, 1.3, 2.0, 3.5, 5.0]))
    e = max(n - 1, int(n * dens))
    its = int(rng.choice([3, 8, 15]))
    xy = bool(rng.random() < 0.25)
    g = synth.make_pose_graph(n, e, seed=int(rng.integers(1, 10**6)), outlier_frac=float(rng.choice([0.0, 0.05, 0.2])))
    kind = rng.choice(["natural", "permuted", "no_odo"], p=[0.6, 0.25, 0.15])
    if kind == "permuted":
        g = synth.permute_graph(g, rng.permutation(n))
    elif kind == "no_odo" and dens >= 2.0:
        g = synth.drop_odometry(g, keep_every=int(rng.choice([0, 4])))
    if ONLY >= 0 and k != ONLY:
        continue
    p.set_config(optimize_xy_only=1 if xy else 0, verbose=1 if ONLY >= 0 else 0)
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    t0 = time.time(); st = p.optimize(its); dt = time.time() - t0
    poses = p.store()[0]
    fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"], optimize_xy_only=xy)
    fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, so = O.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=its)
    d = synth.pose_errors(poses.reshape(-1, 3, 4), P.reshape(-1, 3, 4))
    same_its = st["iterations_done"] == so["iterations_done"] or st["terminated_early"] or so["terminated_early"]
    ok = st["status"] == 0 and d[0] < 1e-3 and d[1] < 1e-4 and same_its
    bad += 0 if ok else 1
    print("%s n %5d e %6d %-8s xy %d its %2d | status %d lm %2d/%2d pcg %6d %.1f ms | dpose %.2e m %.2e rad" %
          ("ok  " if ok else "MISS", n, len(fl["ij"]), kind, xy, its, st["status"], st["iterations_done"], so["iterations_done"], st["pcg_iterations"], 1e3 * dt, d[0], d[1]), flush=True)
p.close()
print("%d cases, %d misses" % (n_cases, bad))
sys.exit(1 if bad else 0)
