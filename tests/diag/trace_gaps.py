#!/usr/bin/env python3
"""Diagnostic: reads a rocprofv3 kernel trace (csv) and reports, for the LAST solve in it (from its last linearize-free stretch back to
the pcg_init of LM iteration 0 is not known here, so: the last `n_lin` linearize_kernel launches), kernel-busy time, idle time, and which
kernels the GPU waited longest BEFORE (host round trips, eager launch gaps).
  python tests/diag/trace_gaps.py kernel_trace.csv [n_lin=20]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uzl::", "")))
rows.sort()
n_lin = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lin = [i for i, r in enumerate(rows) if r[2].startswith("linearize_kernel")]
first = lin[-n_lin]
sel = rows[first:]
span = sel[-1][1] - sel[0][0]
busy = 0; cur_end = sel[0][0]
gap_before = defaultdict(lambda: [0, 0]); ktime = defaultdict(lambda: [0, 0])
for s, e, k in sel:
    if s > cur_end:
        gap_before[k][0] += s - cur_end; gap_before[k][1] += 1
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    ktime[k][0] += e - s; ktime[k][1] += 1
print("last solve: span %.2f ms, kernels busy %.2f ms, idle %.2f ms, %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(sel)))
print("idle time in front of (top 12):")
for k, (g, c) in sorted(gap_before.items(), key=lambda x: -x[1][0])[:12]:
    print("   %-44s %8.3f ms over %5d gaps (%.1f us each)" % (k[:44], g / 1e6, c, g / 1e3 / c))
print("kernel time (top 12):")
for k, (t, c) in sorted(ktime.items(), key=lambda x: -x[1][0])[:12]:
    print("   %-44s %8.3f ms over %5d launches (%.1f us each)" % (k[:44], t / 1e6, c, t / 1e3 / c))
