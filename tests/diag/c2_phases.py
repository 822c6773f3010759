#!/usr/bin/env python3
"""Diagnostic: one BASELINE config-2 graph solved repeatedly; with UZL_PHASES=1 (diagnostic build: UZL_LIB=.../libuzl_mi355x_diag.so) the pass
driver prints the GPU time between its segment boundaries (uzl_pgo_lm.hip)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 5000)
g = synth.make_pose_graph(n, e)
p = capi.Pgo()
p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
p.optimize(20)
ts = []
for _ in range(6):
    p.reset()
    t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("%d/%d: best %.3f ms median %.3f ms  pcg %d trials %d passes %d builds %d" % (n, e, ts.min(), np.median(ts), st["pcg_iterations"], st["lm_trials"], st["lm_passes"], st["precond_builds"]), flush=True)
p.close()
