#!/usr/bin/env python3
"""Diagnostic: how far do the online run's poses (BASELINE config 5) depend on the linear solver's accuracy?  The same run at the default
stop test and at two much tighter ones; per re-optimisation interval the largest pose difference between the runs.  (Every interval starts
from the previous interval's result, so a difference is what twenty LM iterations leave of it plus what the interval adds.)
  python tests/diag/c5_tolerance.py [n_nodes] [n_pairs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import online, synth   # noqa: E402

n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
run = synth.make_online_run(n_nodes, n_pairs, n_kp=1000)
res = {}
for name, cfg in (("default", {}), ("tol 1e-7", dict(pcg_tol=1e-7)), ("relative 1e-9", dict(pcg_stop=1, pcg_tol=1e-9))):
    o = online.OnlineSlam(run, match_batch=512, pgo_cfg=cfg)
    o.keep_poses_per_solve = 1000
    o.upload_frames()
    wall = o.run_all()
    res[name] = (o.poses_at_solve, [s["pcg_iterations"] for s in o.solves], [s["chi2_final"] for s in o.solves], wall, np.array(o.accept_log))
    o.close()
    print("%-14s wall %.2f s, %d PCG iterations" % (name, wall, sum(res[name][1])), flush=True)
ref = res["relative 1e-9"]
for name in ("default", "tol 1e-7"):
    r = res[name]
    assert np.array_equal(r[4], ref[4]), "accepted-edge sets differ"
    d = [synth.pose_errors(a, b) for a, b in zip(r[0], ref[0])]
    dt = np.array([x[0] for x in d]); dr = np.array([x[1] for x in d])
    k = int(np.argmax(dt))
    print("%-14s vs relative 1e-9: largest dt %.2e m at interval %d (dr %.2e rad), median dt %.2e, last interval dt %.2e dr %.2e; intervals with dt > 1e-4: %d"
          % (name, dt.max(), k + 1, dr[k], np.median(dt), dt[-1], dr[-1], int((dt > 1e-4).sum())))
    print("   dt by interval (x 1e-6 m):", " ".join("%d" % round(1e6 * x) for x in dt))
