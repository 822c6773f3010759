#!/usr/bin/env python3
"""Diagnostic: per-launch time of the two PCG kernels of the large-graph path against the number of workgroups (rounds on 256 CUs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth    # noqa: E402

for n in [int(x) for x in sys.argv[1:]] or [4097, 8193, 10000, 12289, 16385, 20000]:
    g = synth.make_pose_graph(n, 5 * n)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    p.optimize(3)
    p.set_profiling(True); p.reset(); st = p.optimize(3); kt = p.kernel_times(); p.set_profiling(False)
    sp, cg = kt["pcg_spmv"], kt["ml_cg"]
    print("n %6d: %4d workgroups  spmv %.2f us  cg %.2f us per launch (%d launches, %d active)" % (n, (n - 1 + 31) // 32, 1e3 * sp["ms"] / sp["launches"], 1e3 * cg["ms"] / cg["launches"], sp["launches"], st["pcg_iterations"]), flush=True)
    p.close()
