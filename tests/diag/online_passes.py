#!/usr/bin/env python3
"""Config 5: passes enqueued against LM trials over the 94 re-optimisations (a pass per trial is the floor; more = solves that outlasted
their pass or stalled for a set-up segment), and the wall clock.   python tests/diag/online_passes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth, online
run = synth.make_online_run(20000, 4096, n_kp=1000)
for rep in range(2):
    o = online.OnlineSlam(run, match_batch=512)
    o.upload_frames()
    t0 = time.perf_counter(); o.run_all(); wall = time.perf_counter() - t0
    s = o.solves
    print("wall %.3f s  optimize %.3f s  passes %d  trials %d  LM iterations %d  pcg %d  (passes per trial %.2f)" % (
        wall, o.t["optimize"], sum(x["lm_passes"] for x in s), sum(x["lm_trials"] for x in s), sum(x["iterations_done"] for x in s),
        sum(x["pcg_iterations"] for x in s), sum(x["lm_passes"] for x in s) / max(1, sum(x["lm_trials"] for x in s))), flush=True)
    o.close()
