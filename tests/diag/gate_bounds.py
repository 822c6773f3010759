#!/usr/bin/env python3
"""Diagnostic: BASELINE config 5 run; for every gate candidate, the straight-line distance and angle between its nodes' current poses,
whether the verdict is decided without the search (gate_decided, gate_kernels.hip), and the path length the search finds."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, online, synth   # noqa: E402

run = synth.make_online_run(20000, 4096, n_kp=300)
o = online.OnlineSlam(run, match_batch=512)
o.upload_frames()
real = o.gate.check
real_set = o.gate.set_graph
cur = {}
rows = []


def keep_set(poses, edges, merged=None):
    cur["poses"] = poses.copy()
    return real_set(poses, edges, merged)


def check(c, want_dist=True):
    t0 = time.perf_counter(); r = real(c, want_dist=True); ms = 1e3 * (time.perf_counter() - t0)
    P = cur["poses"].reshape(-1, 3, 4)
    A = P[c["from"]]; B = P[c["to"]]
    d = np.linalg.norm(B[:, :, 3] - A[:, :, 3], axis=1)
    Rd = np.einsum("nji,njk->nik", A[:, :, :3], B[:, :, :3])
    ang = np.degrees(np.arccos(np.clip((np.trace(Rd, axis1=1, axis2=2) - 1) / 2, -1, 1)))
    for k in range(len(c)):
        rows.append((d[k], ang[k], r[2][k], r[0][k], ms / len(c)))
    return r


o.gate.set_graph = keep_set
o.gate.check = check
o.run_all()
o.close()
R = np.array(rows)
d, ang, dist, acc = R[:, 0], R[:, 1], R[:, 2], R[:, 3]
searched = dist >= 0
ssf = 0.1
decided = (2 * ssf * d * (1 - 1e-9) + 1 > d) & (10 * ssf * d * (1 - 1e-9) + 30 > ang)
print("%d candidates, %d reached the search, %d of those decided by the straight line" % (len(R), searched.sum(), (searched & decided).sum()))
rest = searched & ~decided
print("the rest (%d): straight line %.2f .. %.2f m (median %.2f), angle up to %.1f deg; path length found %.1f .. %.1f m (median %.1f); needed > %.1f m at most"
      % (rest.sum(), d[rest].min(), d[rest].max(), np.median(d[rest]), ang[rest].max(), dist[rest].min(), dist[rest].max(), np.median(dist[rest]),
         (5 * (d[rest] - 1)).max()))
print("ratio path / needed: min %.1f" % (dist[rest] / np.maximum(5 * (d[rest] - 1), 1e-9)).min())
