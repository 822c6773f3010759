#!/usr/bin/env python3
"""Diagnostic: chain-like graphs (few loop closures per node, the shape of an online run) against the oracle's direct solve.
  python tests/diag/sparse_loops.py n_nodes n_edges [iterations]      (A/B switches through the environment, UZL_VERBOSE=1 for the trial log)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O                                   # noqa: E402
from uzliti_slam_amd import capi, synth              # noqa: E402

n, e = int(sys.argv[1]), int(sys.argv[2])
its = int(sys.argv[3]) if len(sys.argv) > 3 else 10
g = synth.make_pose_graph(n, e)
p = capi.Pgo(preconditioner=int(os.environ.get("PRECOND", "1")))
p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
t0 = time.time(); st = p.optimize(its); dt = time.time() - t0
poses = p.store()[0]
if os.environ.get("NO_ORACLE"):
    print("n %d e %d its %d | gpu: status %d its %d trials %d pcg %d chi2 %.6g -> %.9g  %.1f ms" % (n, e, its, st["status"], st["iterations_done"], st["lm_trials"], st["pcg_iterations"], st["chi2_initial"], st["chi2_final"], 1e3 * dt))
    sys.exit(0)
fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
t0 = time.time()
P, so = O.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=its)
dto = time.time() - t0
d = synth.pose_errors(poses.reshape(-1, 3, 4), P.reshape(-1, 3, 4))
print("n %d e %d its %d | gpu: status %d its %d trials %d pcg %d chi2 %.6g -> %.6g  %.1f ms | oracle: its %d trials %d chi2 %.6g -> %.6g %.1f s | dpose %.3e m %.3e rad"
      % (n, e, its, st["status"], st["iterations_done"], st["lm_trials"], st["pcg_iterations"], st["chi2_initial"], st["chi2_final"], 1e3 * dt,
         so["iterations_done"], so["lm_trials"], so["chi2_initial"], so["chi2_final"], dto, d[0], d[1]))
