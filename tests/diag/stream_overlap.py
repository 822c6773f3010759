#!/usr/bin/env python3
"""Diagnostic: which of n freshly made streams stand in each other's way (uzl_debug_stream_pairs: the stream pool's own decision, one
measurement per unordered pair - two chains of 5 dependent 7-us kernels side by side against one chain alone, timed on the device,
best of 3, in percent)?  ~104 - 148: independent; ~202: one hardware queue; ~250 - 260: two queues in each other's way.
python tests/diag/stream_overlap.py [n] [priority | 200 = priorities 0 and -1 in turn] [repeats]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = capi.diag_lib()
P32 = ctypes.POINTER(ctypes.c_int32)
lo, hi = [], []
for _ in range(reps):
    v = np.zeros((n, n), np.int32)
    r = np.zeros((n, n), np.int32)
    ms = ctypes.c_double(0.)
    rc = lib.uzl_debug_stream_pairs(ctypes.c_int(n), ctypes.c_int(prio), v.ctypes.data_as(P32), r.ctypes.data_as(P32), ctypes.byref(ms))
    print("rc", rc, "priority", prio, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "probe_ms %.2f for %d pairs" % (ms.value, n * (n - 1) // 2))
    for i in range(n):
        print(" ".join("    ." if x < 0 else "%5d" % x for x in r[i]), "  |  ", " ".join("." if x < 0 else str(x) for x in v[i]))
    off = ~np.eye(n, dtype=bool)
    lo += list(r[off & (v == 1)]); hi += list(r[off & (v == 0)])
print("independent pairs: %d, ratio %s .. %s; colliding pairs: %d, ratio %s .. %s" % (
    len(lo) // 2, min(lo, default=None), max(lo, default=None), len(hi) // 2, min(hi, default=None), max(hi, default=None)))
