#!/usr/bin/env python3
"""Diagnostic: the Schur-reduced solve against the full-system solve of the same graph (uzl_pgo_cfg::schur_reduce = 0 / -1).
  python tests/diag/schur_ab.py N:E[:its] ...     (default: a ladder of chain-like graphs)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth   # noqa: E402

cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(2000, 2040), (5000, 5400), (10000, 10900), (20000, 21800), (20000, 23600), (20000, 30000)]
for c in cases:
    n, e = c[0], c[1]
    its = c[2] if len(c) > 2 else 20
    g = synth.make_pose_graph(n, e, seed=n + 1)
    row = "%6d/%6d its %2d:" % (n, e, its)
    ref = None
    for mode in (-1, 0):
        p = capi.Pgo(schur_reduce=mode)
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        p.optimize(its)
        best = 1e30
        for _ in range(3):
            p.reset(); t0 = time.perf_counter(); st = p.optimize(its); best = min(best, time.perf_counter() - t0)
        poses = p.store()[0].reshape(-1, 3, 4)
        if ref is None:
            ref = poses
        dt, dr = synth.pose_errors(poses, ref)
        row += "  %s %8.2f ms pcg %5d trials %2d elim %5d chi2 %.6g" % ("full   " if mode < 0 else "reduced", 1e3 * best, st["pcg_iterations"], st["lm_trials"],
                                                                         st["n_eliminated"], st["chi2_final"])
        if mode == 0:
            row += "  (vs full: %.1e m %.1e rad)" % (dt, dr)
        if len(sys.argv) > 1 and mode == 0:
            p.set_profiling(True); p.reset(); p.optimize(its)
            kt = p.kernel_times()
            row += "\n      " + "  ".join("%s %.3f/%d" % (k, v["ms"], v["launches"]) for k, v in sorted(kt.items(), key=lambda x: -x[1]["ms"])[:12])
        p.close()
    print(row, flush=True)
