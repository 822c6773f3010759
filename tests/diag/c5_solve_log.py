#!/usr/bin/env python3
"""Diagnostic: one solve of config 5's last re-optimisation input (tests/diag/data/c5_last.npz) with the verbose LM log, then timing."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi   # noqa: E402

_here = os.path.dirname(os.path.abspath(__file__))
_cands = [os.path.join(_here, "data", "c5_last.npz"), os.path.join(os.path.dirname(os.path.dirname(_here)), "gpurun_out", "c5_last.npz")]
_found = [f for f in _cands if os.path.exists(f)]
if not _found:
    sys.exit("no c5_last.npz: run tests/diag/c5_last.py on a GPU box first (it writes gpurun_out/c5_last.npz; 4 MB, not tracked)")
z = np.load(_found[0])
e = {k[2:]: z[k] for k in z.files if k.startswith("e_")}
p = capi.Pgo(**({"lm_loop": int(os.environ["LM_LOOP"])} if "LM_LOOP" in os.environ else {}))
p.add_graph(z["poses"], z["fixed"], e); p.optimize(20)
ts = []
for _ in range(5):
    p.add_graph(z["poses"], z["fixed"], e); t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
print("solve: best %.2f ms median %.2f ms" % (1e3 * min(ts), 1e3 * sorted(ts)[2]), {k: st[k] for k in ("iterations_done", "lm_trials", "pcg_iterations", "n_eliminated", "lm_passes")}, flush=True)
if os.environ.get("LOG", "1") != "0":
    p.set_config(verbose=1)
    p.add_graph(z["poses"], z["fixed"], e); p.optimize(20)
    p.set_config(verbose=0)
p.set_profiling(True); p.add_graph(z["poses"], z["fixed"], e); p.optimize(20)
for k, v in sorted(p.kernel_times().items(), key=lambda x: -x[1]["ms"])[:14]:
    print("  %-18s %8.3f ms %6d launches %7.2f us" % (k, v["ms"], v["launches"], 1e3 * v["ms"] / max(v["launches"], 1)))
p.close()
