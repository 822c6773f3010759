#!/usr/bin/env python3
"""Diagnostic: reads a rocprofv3 kernel trace (csv) and reports, over the WHOLE trace and within a time window of activity, where the GPU
idled: total busy / idle time (gaps longer than `long_ms` are taken to be the host doing something else and listed apart) and the idle
time in front of each kernel name.
  python tests/diag/trace_idle.py kernel_trace.csv [long_ms=1.0]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uzl::", "")))
rows.sort()
long_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e6
busy = 0; cur_end = rows[0][0]
gap_before = defaultdict(lambda: [0, 0]); long_gaps = defaultdict(lambda: [0, 0])
for s, e, k in rows:
    if s > cur_end:
        g = s - cur_end
        tgt = long_gaps if g > long_ns else gap_before
        tgt[k][0] += g; tgt[k][1] += 1
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
span = rows[-1][1] - rows[0][0]
short = sum(v[0] for v in gap_before.values()); lng = sum(v[0] for v in long_gaps.values())
print("span %.1f ms: kernels busy %.1f ms, short gaps %.1f ms, long gaps (> %.1f ms each) %.1f ms, %d launches" % (span / 1e6, busy / 1e6, short / 1e6, long_ns / 1e6, lng / 1e6, len(rows)))
print("short idle time in front of (top 14):")
for k, (g, c) in sorted(gap_before.items(), key=lambda x: -x[1][0])[:14]:
    print("   %-46s %8.3f ms over %6d gaps (%.1f us each)" % (k[:46], g / 1e6, c, g / 1e3 / c))
print("long gaps in front of (top 8):")
for k, (g, c) in sorted(long_gaps.items(), key=lambda x: -x[1][0])[:8]:
    print("   %-46s %8.3f ms over %6d gaps (%.2f ms each)" % (k[:46], g / 1e6, c, g / 1e6 / c))
