#!/usr/bin/env python3
"""Diagnostic: per-trial PCG log of one solve (UZL_VERBOSE) and the deviation from the CPU checker after every LM iteration count.
  python tests/diag/lm_trace.py N E [its] [npz of a saved graph instead]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["UZL_VERBOSE"] = "1"
import oracle as O                                   # noqa: E402
from uzliti_slam_amd import capi, synth              # noqa: E402

n, e = int(sys.argv[1]), int(sys.argv[2])
its = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if len(sys.argv) > 4:
    z = np.load(sys.argv[4])
    g = dict(nodes_pose=z["poses"], nodes_fixed=z["fixed"], edges={k[2:]: z[k] for k in z.files if k.startswith("e_")})
else:
    g = synth.make_pose_graph(n, e)
fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
p = capi.Pgo()
for k in ([1, 2, 3, 5, 8, 12, 20] if its == 20 else [its]):
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st = p.optimize(k)
    poses = p.store()[0]
    P, so = O.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=k)
    dt, dr = synth.pose_errors(poses.reshape(-1, 3, 4), P.reshape(-1, 3, 4))
    print("after %2d LM iterations: pcg %5d  trials %d/%d  dpose %.2e m %.2e rad  chi2 %.9g / %.9g" % (k, st["pcg_iterations"], st["lm_trials"], so["lm_trials"], dt, dr,
                                                                                                  st["chi2_final"], so["chi2_final"]), flush=True)
p.close()
