#!/usr/bin/env python3
"""Diagnostic: which of n freshly made streams stand in each other's way (uzl_debug_stream_pairs: two chains of 32 dependent ~4-us kernels
on streams i and j at once against one chain on stream i, in percent)?  ~105 - 125: independent; ~200: one hardware queue; ~270: two
queues on one compute pipe.   python tests/diag/stream_overlap.py [n] [priority | 200 = priorities 0 and -1 in turn]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = capi.diag_lib()
out = np.zeros((n, n), np.int32)
rc = lib.uzl_debug_stream_pairs(ctypes.c_int(n), ctypes.c_int(prio), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
print("rc", rc, "priority", prio, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for i in range(n):
    print(" ".join("    ." if v < 0 else "%5d" % v for v in out[i]))
