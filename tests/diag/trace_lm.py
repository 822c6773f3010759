#!/usr/bin/env python3
"""Diagnostic: timeline of the LAST solve in a rocprofv3 kernel trace (csv): per kernel name total time / count / mean, GPU busy vs idle,
and the idle time in front of each kernel name.   python tests/diag/trace_lm.py kernel_trace.csv [n_lin=20]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uzl::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
rows.sort()
n_lin = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lin = [i for i, r in enumerate(rows) if r[2].startswith("linearize")]
# the last solve: from the n_lin-th last ACTIVE linearize (duration > 2 us means it did work)
act = [i for i in lin if rows[i][1] - rows[i][0] > 2500]
first = act[-n_lin]
sel = rows[first:]
span = sel[-1][1] - sel[0][0]
busy = 0; cur_end = sel[0][0]
gap = defaultdict(lambda: [0, 0]); kt = defaultdict(lambda: [0, 0])
for s, e, k in sel:
    if s > cur_end:
        gap[k][0] += s - cur_end; gap[k][1] += 1
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    kt[k][0] += e - s; kt[k][1] += 1
print("last solve: span %.3f ms, busy %.3f ms, idle %.3f ms, %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(sel)))
print("kernel time:")
for k, (t, c) in sorted(kt.items(), key=lambda x: -x[1][0])[:25]:
    print("   %-60s %8.3f ms %5d x %6.2f us" % (k[:60], t / 1e6, c, t / 1e3 / c))
print("idle in front of:")
for k, (g, c) in sorted(gap.items(), key=lambda x: -x[1][0])[:15]:
    print("   %-60s %8.3f ms %5d x %6.2f us" % (k[:60], g / 1e6, c, g / 1e3 / c))
