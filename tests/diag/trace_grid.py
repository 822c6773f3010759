#!/usr/bin/env python3
"""Diagnostic: per (kernel name, grid) statistics of a rocprofv3 kernel trace, restricted to kernels whose name contains one of the
given substrings.   python tests/diag/trace_grid.py kernel_trace.csv [substr ...]"""
import csv
import sys
from collections import defaultdict

subs = sys.argv[2:] or ["ml_"]
st = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uzl::", "")
    if not any(s in nm for s in subs):
        continue
    wg = int(r["Workgroup_Size_X"])
    key = (nm[:44], int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), wg)
    st[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in sorted(st.items(), key=lambda x: -sum(x[1])):
    v = sorted(v)
    tot += sum(v)
    print("%-44s wgs %6d x%3d x%3d (%4d)  %5d x  min %7.2f  med %7.2f  max %7.2f us   total %9.1f us" % (k + (len(v), v[0] / 1e3, v[len(v) // 2] / 1e3, v[-1] / 1e3, sum(v) / 1e3)))
print("total %.1f us" % (tot / 1e3))
