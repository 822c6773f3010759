#!/usr/bin/env python3
"""Diagnostic: BASELINE config 5 run; how many gate searches ran on the wave kernel / fell back to the lane kernel, gate time per call."""
import ctypes as C
import os

import numpy as np
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, online, synth   # noqa: E402

run = synth.make_online_run(20000, 4096, n_kp=300)
o = online.OnlineSlam(run, match_batch=512)
o.upload_frames()
calls = []
real = o.gate.check


saved = {}
real_set = o.gate.set_graph


def keep_set(poses, edges, merged=None):
    saved["cur"] = (poses.copy(), edges.copy())
    return real_set(poses, edges, merged)


def timed_check(c):
    t0 = time.perf_counter(); r = real(c); ms = 1e3 * (time.perf_counter() - t0)
    if os.environ.get("UZL_GATE_DBG") == "1" and ms > saved.get("ms", 0.0):
        buf = np.zeros(8 * len(c), np.int64)
        nq = capi.lib().uzl_debug_gate_profile(o.gate._h, buf.ctypes.data_as(C.c_void_p), C.c_int32(len(c)))
        saved["prof"] = buf[:8 * nq].reshape(-1, 8).copy()
    calls.append((len(c), ms, float(r[2].max(initial=0.0))))
    if ms > saved.get("ms", 0.0):
        saved["ms"] = ms; saved["graph"] = saved["cur"]; saved["cand"] = c.copy()
    return r


o.gate.set_graph = keep_set
o.gate.check = timed_check
o.run_all()
nw = C.c_int64(); nl = C.c_int64()
capi.lib().uzl_gate_search_counts(o.gate._h, C.byref(nw), C.byref(nl))
print("searches: wave/lds kernel %d, lane-kernel fallback %d; gate %.3f s over %d calls" % (nw.value, nl.value, o.t["gate"], len(calls)))
for k in range(0, len(calls), 8):
    print("  call %3d: %3d candidates %7.2f ms  longest path %.1f m" % (k, calls[k][0], calls[k][1], calls[k][2]))
o.close()

# ---- the slowest call again: expansions per search (CPU checker's counter) against the kernel's time
import numpy as np   # noqa: E402
import oracle as O   # noqa: E402
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/gate_slowest.npz", poses=saved["graph"][0], edges=saved["graph"][1], cand=saved["cand"])
g = O.Gate()
g.set_graph(*saved["graph"])
ex = []
t0 = time.perf_counter()
for c in saved["cand"]:
    g.astar(int(c["from"]), int(c["to"])); ex.append(g.last_expansions())
cpu_ms = 1e3 * (time.perf_counter() - t0)
ex = np.array(ex)
print("slowest call: %.2f ms on the GPU for %d candidates; expansions per search: max %d, mean %.0f, total %d; CPU checker, one after the other: %.1f ms"
      % (saved["ms"], len(ex), ex.max(), ex.mean(), ex.sum(), cpu_ms))
print("=> %.2f us per expansion of the longest search" % (1e3 * saved["ms"] / ex.max()))
if "prof" in saved:
    pr = saved["prof"]; i = int(pr[:, 0].argmax())
    print("kernel counters of that call's longest search: %d steps, %.0f shader clocks per step, %.2f us per step (100 MHz counter), largest list %d"
          % (pr[i, 0], pr[i, 1] / max(pr[i, 0], 1), 1e-2 * pr[i, 2] / max(pr[i, 0], 1), pr[i, 3]))
    print("   clocks per step: pop %.0f, popped node %.0f, neighbours %.0f, pushes %.0f" % tuple(pr[i, 4:8] / max(pr[i, 0], 1)))
