import sys, os, ctypes as C, time
sys.path.insert(0, os.getcwd())
import numpy as np
from uzliti_slam_amd import capi, synth
# config 2 / config 4 profiled solves: GEMM kernel times through the handle's kernel timer
for n, e in ((1000, 5000), (10000, 50000), (2500, 12000), (5000, 25000)):
    g = synth.make_pose_graph(n, e, seed=12345)
    p = capi.Pgo(pass_history=1)
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20)
    t0 = time.perf_counter()
    for _ in range(5):
        p.reset(); st = p.optimize(20)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    p.set_profiling(True); p.reset(); st = p.optimize(20); kt = p.kernel_times(); p.set_profiling(False)
    gm = kt.get("ml_ns_gemm")
    print("%d/%d: solve %.3f ms, pcg %d; ml_ns_gemm %d launches x %.1f us" % (n, e, ms, st["pcg_iterations"], gm["launches"], 1e3 * gm["ms"] / gm["launches"]))
    p.close()
