#!/usr/bin/env python3
"""knn2 on 64-byte descriptors (BRISK-512): kernel time at the deployed point (300 keypoints) and at 1000 keypoints.
   UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so UZL_KNN2_RB16=1|2 python tests/diag/knn16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from uzliti_slam_amd import capi, synth
for nkp in (300, 1000):
    pairs = synth.make_pairs(512, n_kp=nkp, desc_bytes=64, seed=4242)
    m = capi.Match(ransac_threshold=0.1, ransac_iteration=100, ransac_break_percentage=0.6, seed=777)
    ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
    jobs, fids = capi.Match._jobs(ids, None)
    res = np.zeros(512, capi.EDGE_RESULT_DTYPE)
    for _ in range(3):
        m.launch_raw(jobs, fids); m.collect(res)
    m.set_profiling(True)
    best = 1e9
    for _ in range(5):
        m.launch_raw(jobs, fids); m.collect(res); best = min(best, m.kernel_times()["knn2"]["ms"])
    ops = 2.0 * 512 * nkp * nkp * 512
    print("RB16=%s  %4d keypoints: knn2 %.4f ms = %.2f POP/s = %.3f of the int8 peak; consensus sum %d" % (os.environ.get("UZL_KNN2_RB16", "default"), nkp, best, ops / best / 1e12 * 1e-0 / 1e0 / 1e0 * 1e-0 if False else ops / (best * 1e-3) / 1e15, ops / (best * 1e-3) / 1e15 / 5.0, int(res["consensus"].sum())))
    m.close()
