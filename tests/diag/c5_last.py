#!/usr/bin/env python3
"""Diagnostic: run BASELINE config 5 to the end, save the input of its last re-optimisation (gpurun_out/c5_last.npz) and profile a
solve of it: per-trial PCG log (UZL_VERBOSE) on stderr, kernel times on stdout."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, online, synth   # noqa: E402

n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
run = synth.make_online_run(n_nodes, n_pairs, n_kp=1000 if n_nodes >= 20000 else 300)
o = online.OnlineSlam(run, match_batch=512)
o.upload_frames()
o.run_all()
poses, fixed, e = o.last_input
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/c5_last.npz", poses=poses, fixed=fixed, **{"e_" + k: np.asarray(v) for k, v in e.items()})
print("solves", len(o.solves), "last", {k: o.solves[-1][k] for k in ("n_nodes", "n_edges", "pcg_iterations", "lm_trials", "optimize_ms", "n_eliminated")})
o.close()
os.environ["UZL_VERBOSE"] = "1"
p = capi.Pgo()
p.add_graph(poses, fixed, e)
p.optimize(20)
for _ in range(2):
    p.add_graph(poses, fixed, e); t0 = time.perf_counter(); st = p.optimize(20); dt = time.perf_counter() - t0
print("fresh solve %.2f ms" % (1e3 * dt), st)
p.set_profiling(True); p.add_graph(poses, fixed, e); p.optimize(20)
for k, v in sorted(p.kernel_times().items(), key=lambda x: -x[1]["ms"]):
    print("  %-18s %8.3f ms %6d launches %7.2f us" % (k, v["ms"], v["launches"], 1e3 * v["ms"] / max(v["launches"], 1)))
p.close()
