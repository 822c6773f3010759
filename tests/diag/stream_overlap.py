#!/usr/bin/env python3
"""Diagnostic: which of n freshly made streams overlap (uzl_debug_stream_overlap: a 200-us wait on stream i, an empty kernel on stream j)?
The runtime serves its streams from a few hardware queues (GPU_MAX_HW_QUEUES, default 4); streams on one queue run one behind the other."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = capi.lib()
out = np.zeros((n, n), np.int32)
rc = lib.uzl_debug_stream_overlap(ctypes.c_int(n), ctypes.c_int(prio), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
print("rc", rc, "priority", prio, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for i in range(n):
    print(" ".join("." if v < 0 else ("o" if v else "X") for v in out[i]))
