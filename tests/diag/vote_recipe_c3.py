#!/usr/bin/env python3
"""All 512 pairs of BASELINE config 3: votes of every hypothesis under the fused recipe (oracle = HIP kernels) and under the reference
build's unfused evaluation order; prints the counts quoted in DESIGN.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O                         # noqa: E402
from uzliti_slam_amd import synth          # noqa: E402

pairs = synth.make_pairs(512, n_kp=1000, seed=777)
tests = diffs = 0
margin = 1e300
for j, (f, t, _) in enumerate(pairs):
    a = O.estimate_edge([f], [t], ransac_threshold=0.1, ransac_iteration=500, break_percentage=1.0, do_prosac=True, seed=777, job_id=j)
    nt, nd, mm = O.vote_recipe_diff(t["pos"][:, a["corr_query"]], f["pos"][:, a["corr_train"]], 0.1, 500, True, 777, j)
    tests += nt; diffs += nd; margin = min(margin, mm)
print("config 3: %d votes, %d differ between the recipes; closest point to the threshold: %.3e m" % (tests, diffs, margin))
