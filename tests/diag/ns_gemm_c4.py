#!/usr/bin/env python3
"""ml_ns_gemm_kernel at config-4 size (n = 1878): per-launch time from a profiled solve (dispatch timestamps), flop rate against the f64
matrix-core peak.  Under `rocprofv3 --pmc FETCH_SIZE` / WRITE_SIZE the same run gives the kernel's HBM-side traffic.
   python tests/diag/ns_gemm_c4.py [nodes edges]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth
n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 50000)
g = synth.make_pose_graph(n, e, seed=12345)
p = capi.Pgo()
p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20)
p.set_profiling(True); p.reset(); st = p.optimize(20); kt = p.kernel_times()
gm = kt["ml_ns_gemm"]
nb = st["n_vertices"] - int(p.get_fixed().sum())
n1 = (nb + 7) // 8; nc = n1 if nb <= 2048 else (n1 + 3) // 4
n6 = 6 * nc; gt = (n6 + 63) // 64
ext = [min(64, n6 - 64 * i) for i in range(gt)]
flop = 2.0 * n6 * sum(ext[i] * ext[j] for i in range(gt) for j in range(i, gt))
us = 1e3 * gm["ms"] / gm["launches"]
print("n = %d: ml_ns_gemm %d launches, %.1f us each, %.1f TFLOP/s = %.3f of 78.6; per solve %.2f ms; solve %.2f ms, %d PCG iterations"
      % (n6, gm["launches"], us, flop / us / 1e6, flop / us / 1e6 / 78.6, gm["ms"], st["solve_ms"], st["pcg_iterations"]))
for k, v in sorted(kt.items(), key=lambda x: -x[1]["ms"])[:12]:
    print("  %-24s %8.3f ms %5d launches" % (k, v["ms"], v["launches"]))
