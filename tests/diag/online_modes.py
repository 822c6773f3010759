#!/usr/bin/env python3
"""Diagnostic: config 5, per re-optimisation: numbering of the reduced system the handle chose, PCG iterations per LM trial, time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import online, synth   # noqa: E402

run = synth.make_online_run(20000, 4096, n_kp=1000)
o = online.OnlineSlam(run, match_batch=512, pgo_cfg=dict(reduced_numbering=int(os.environ.get("NUMBERING", "0"))))
o.upload_frames()
wall = o.run_all()
line = []
for k, s in enumerate(o.solves):
    line.append("%d:%s%.0f/%.1f" % (k + 1, "S" if s.get("reduced_strong") else ("r" if s.get("n_eliminated", 0) else "-"), s["pcg_iterations"] / max(s["lm_trials"], 1), s["optimize_ms"]))
print(" ".join(line))
print("wall %.3f s, pcg %d" % (wall, sum(s["pcg_iterations"] for s in o.solves)))
o.close()
