#!/usr/bin/env python3
"""Diagnostic: one online run (BASELINE config 5) on cuda:0, per-solve log on stderr, summary JSON on stdout.
  python tests/diag/online_run.py [n_nodes] [n_pairs] [n_kp] [match_batch]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import online, synth   # noqa: E402

n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_kp = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 512
t0 = time.time()
run = synth.make_online_run(n_nodes, n_pairs, n_kp=n_kp)
print("synth %.1f s" % (time.time() - t0), file=sys.stderr)
o = online.OnlineSlam(run, match_batch=batch, log=lambda m: print(m, file=sys.stderr))
o.upload_frames()
wall = o.run_all()
s = o.summary(wall)
import numpy as np   # noqa: E402
gt = run["gt"]; est = o.poses
s["ate_dead_reckoning_m"] = float(np.linalg.norm(run["init"][:, :, 3] - gt[:, :, 3], axis=1).mean())
s["ate_online_m"] = float(np.linalg.norm(est[:, :, 3] - gt[:, :, 3], axis=1).mean())
acc = np.array(o.accept_log)
s["gate_candidates"] = int(len(acc)); s["gate_accepted"] = int(acc[:, 1].sum()) if len(acc) else 0
alias = run["pair_alias"]
s["aliased_pairs"] = int(alias.sum()); s["aliased_accepted"] = int(alias[o.f_key].sum()); s["aliased_valid"] = int(alias[o.f_key[o.f_sticky]].sum())
print(json.dumps(s))
o.close()
