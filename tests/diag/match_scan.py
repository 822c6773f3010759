#!/usr/bin/env python3
"""Diagnostic: the estimator's time over frame sizes and descriptor widths (512 pairs, 500 iterations, no early exit) - a scan for cliffs
like the solver's tests/diag/big_graphs.py.   python tests/diag/match_scan.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

for desc_bytes in (32, 64):
    for n_kp in (100, 300, 500, 1000, 1500, 2000, 3000, 4096, 5000, 8000):
        n_pairs = 512 if n_kp <= 2000 else 128
        pairs = synth.make_pairs(n_pairs, n_kp=n_kp, desc_bytes=desc_bytes, seed=99)
        m = capi.Match(ransac_threshold=0.1, ransac_iteration=500, ransac_break_percentage=1.0, do_prosac=1, seed=777)
        ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
        jobs, fids = capi.Match._jobs(ids, None)
        res = np.zeros(n_pairs, capi.EDGE_RESULT_DTYPE)
        for _ in range(3):
            m.launch_raw(jobs, fids); m.collect(res)
        t0 = time.perf_counter()
        for _ in range(10):
            m.launch_raw(jobs, fids); m.collect(res)
        dt = (time.perf_counter() - t0) / 10
        m.set_profiling(True); m.launch_raw(jobs, fids); m.collect(res); mk = m.kernel_times(); m.set_profiling(False)
        print("%2d B x %5d keypoints, %3d pairs: %8.3f ms = %9.0f pairs/s   (%s)  mean correspondences %.0f  ok %d" % (
            desc_bytes, n_kp, n_pairs, 1e3 * dt, n_pairs / dt, ", ".join("%s %.3f" % (k, v["ms"]) for k, v in mk.items()), float(res["n_corr"].mean()), int(res["ok"].sum())), flush=True)
        m.close()
