#!/usr/bin/env python3
"""optimize again FROM the solved poses (the reference's timer tick on an unchanged graph): what one such call costs and why.
   UZL_VERBOSE=1 python tests/diag/continued.py 2>&1 | tail -40"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth
g = synth.make_pose_graph(1000, 5000, seed=12345)
p = capi.Pgo(pass_history=1, verbose=0)
p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20); p.reset(); p.optimize(20)
for k in range(4):
    if k == 3: p.set_config(verbose=1)
    t0 = time.perf_counter(); st = p.optimize(20); dt = time.perf_counter() - t0
    sys.stderr.write("[diag] call %d: %.3f ms wall, solve_ms %.3f structure_ms %.3f, %d LM iterations, %d trials, %d passes, %d pcg, terminated %d chi2 %.6f\n"
                     % (k, 1e3 * dt, st["solve_ms"], st["structure_ms"], st["iterations_done"], st["lm_trials"], st["lm_passes"], st["pcg_iterations"], st["terminated_early"], st["chi2_final"]))
