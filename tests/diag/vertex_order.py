#!/usr/bin/env python3
"""Diagnostic: PCG iteration counts when the vertex index order is not the trajectory order (permuted ids, two interleaved sessions,
no odometry chain), each against the oracle's direct solve."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O                                   # noqa: E402
from uzliti_slam_amd import capi, synth              # noqa: E402


def run(name, g, its=10):
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    t0 = time.time(); st = p.optimize(its); dt = time.time() - t0
    poses = p.store()[0]
    p.close()
    fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, so = O.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=its)
    d = synth.pose_errors(poses.reshape(-1, 3, 4), P.reshape(-1, 3, 4))
    print("%-34s n %5d e %6d | status %d its %d pcg %6d (%.0f / LM it) %.1f ms chi2 %.6g | oracle its %d chi2 %.6g | dpose %.2e m %.2e rad"
          % (name, len(g["nodes_fixed"]), len(g["edges"]["from"]), st["status"], st["iterations_done"], st["pcg_iterations"],
             st["pcg_iterations"] / max(st["lm_trials"], 1), 1e3 * dt, st["chi2_final"], so["iterations_done"], so["chi2_final"], d[0], d[1]), flush=True)


for n, e in ((1000, 5000), (4000, 20000)):
    g = synth.make_pose_graph(n, e)
    run("natural order", g)
    rng = np.random.default_rng(1)
    run("random permutation", synth.permute_graph(g, rng.permutation(n)))
    run("reversed", synth.permute_graph(g, np.arange(n)[::-1].copy()))
    run("two interleaved sessions", synth.interleave_sessions(synth.make_pose_graph(n // 2, e // 2, seed=1), synth.make_pose_graph(n // 2, e // 2, seed=2)))
    run("no odometry chain", synth.drop_odometry(g))
    run("every 5th odometry edge", synth.drop_odometry(g, keep_every=5))
