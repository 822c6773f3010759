#!/usr/bin/env python3
"""Diagnostic: PCG iterations and time of synthetic chain-like graphs with the reduced system in row order, numbered by strong aggregates,
and with the handle choosing (uzl_pgo_cfg::reduced_numbering = 1 / 2 / 0).   python tests/diag/strong_ab.py"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from uzliti_slam_amd import capi, synth   # noqa: E402
    for n, e in ((3000, 3100), (8000, 8400), (8000, 9000), (12000, 12700), (20000, 21700), (20000, 24000)):
        g = synth.make_pose_graph(n, e, seed=n + e)
        p = capi.Pgo(reduced_numbering=int(os.environ["NUMBERING"]))
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20)
        ts = []
        for _ in range(3):
            p.reset(); t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
        print("  %5d/%5d: %7.2f ms  pcg %5d  trials %d  eliminated %d  strong %d  chi2 %.6g" % (n, e, 1e3 * min(ts), st["pcg_iterations"], st["lm_trials"], st["n_eliminated"], st["reduced_strong"], st["chi2_final"]), flush=True)
        p.close()
else:
    for tag, env in (("row order", {"NUMBERING": "1"}), ("strong aggregates", {"NUMBERING": "2"}), ("the handle chooses", {"NUMBERING": "0"})):
        print(tag, flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
