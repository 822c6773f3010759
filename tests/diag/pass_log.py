#!/usr/bin/env python3
"""Per-pass log of one solve (UZL_VERBOSE=1): what the host predicted (PCG iterations enqueued) against what the solve took.
   UZL_VERBOSE=1 python tests/diag/pass_log.py nodes edges [pass_history] 2>&1 | grep -E "pass|loop:" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth
n, e = int(sys.argv[1]), int(sys.argv[2])
hist = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = synth.make_pose_graph(n, e, seed=12345 if (n, e) == (1000, 5000) else 4040)
p = capi.Pgo(pass_history=hist, verbose=0)
p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20)
p.reset()
p.set_config(verbose=1)
st = p.optimize(20)
sys.stderr.write("[diag] %d passes, %d trials, %d pcg iterations, %.3f ms\n" % (st["lm_passes"], st["lm_trials"], st["pcg_iterations"], st["solve_ms"]))
