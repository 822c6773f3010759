#!/usr/bin/env python3
"""What handles cost to make: uzl_pgo_create (two streams, two events), uzl_pgo_batch_create (n handles + the batch's streams from the pool),
first and second batch of a process.   python tests/diag/create_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi
capi.lib()
t0 = time.perf_counter(); p = capi.Pgo(); t1 = time.perf_counter()
print("first uzl_pgo_create (runtime initialisation included): %.2f ms" % (1e3 * (t1 - t0)))
hs = []
t0 = time.perf_counter()
for _ in range(32):
    hs.append(capi.Pgo())
t1 = time.perf_counter()
print("uzl_pgo_create: %.3f ms each (32 handles)" % (1e3 * (t1 - t0) / 32))
t0 = time.perf_counter()
for h in hs:
    h.close()
print("uzl_pgo_destroy: %.3f ms each" % (1e3 * (time.perf_counter() - t0) / 32))
for rnd in range(3):
    for n in (4, 16, 64):
        t0 = time.perf_counter(); b = capi.PgoBatch(n); t1 = time.perf_counter()
        st = capi.stream_stats(0)
        b.close(); t2 = time.perf_counter()
        print("round %d: uzl_pgo_batch_create(%2d) %.2f ms, destroy %.2f ms; pool %s" % (rnd, n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), st))
