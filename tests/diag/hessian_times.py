#!/usr/bin/env python3
"""hessian_kernel's dispatch time in a profiled solve (config 2 and 10k / 50k), with the roofline fraction of SURVEY 8(d)'s 632 E + 336 N bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth
for n, e in ((1000, 5000), (10000, 50000), (20000, 24000)):
    g = synth.make_pose_graph(n, e, seed=12345)
    p = capi.Pgo(pass_history=1)
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20)
    p.set_profiling(True); p.reset(); st = p.optimize(20); kt = p.kernel_times(); p.set_profiling(False)
    lin = kt["linearize"]; us = 1e3 * lin["ms"] / lin["launches"]
    alg = 632.0 * st["n_edges"] + 336.0 * st["n_vertices"]
    print("%d/%d: hessian_kernel %d launches x %.2f us = %.3f of 8 TB/s; chi2 %.9g -> %.9g, pcg %d" % (n, e, lin["launches"], us, alg / us / 1e6 / 8., st["chi2_initial"], st["chi2_final"], st["pcg_iterations"]))
    p.close()
