#!/usr/bin/env python3
"""Condenses a profiles/collect.sh run into small committed files:
   <out>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats, verbatim)
   <out>_pmc.json          (per kernel: launches, mean FETCH_SIZE / WRITE_SIZE in KB, HBM-side bytes per launch with
                            the gfx950 correction FETCH x2 for 16-B/lane streaming reads, MI355X_MICROARCH.md §HBM)
   profiles/traffic.json   (bytes per launch of the two roofline kernels; read by bench.py)"""
import collections
import csv
import glob
import json
import shutil
import sys

src, out = sys.argv[1], sys.argv[2]
import os
ks = sorted(glob.glob(f"{src}/trace/*/*kernel_stats.csv"), key=os.path.getmtime, reverse=True)
if ks:
    shutil.copy(ks[0], out + "_kernel_stats.csv")
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob(f"{src}/{c}/*/*counter_collection.csv"), key=os.path.getmtime, reverse=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c:
            continue
        v = float(r["Counter_Value"])
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        # launches that exit at the `done` flag move (almost) nothing: keep them out of the per-launch mean
        if v < 1.0 and ("ml_" in k or "pcg_" in k or "_lm_kernel" in k):      # (slot twins no-op unless their graph is in the phase they serve)
            continue
        agg[k][0] += 1
        agg[k][1] += v
    for k, (n, v) in agg.items():
        pmc.setdefault(k, {})[c] = dict(launches=n, mean_kb=v / max(n, 1))
for k, d in pmc.items():
    f = d.get("FETCH_SIZE", {}).get("mean_kb", 0.0)
    w = d.get("WRITE_SIZE", {}).get("mean_kb", 0.0)
    d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
# the 10k/50k run: its own kernel stats; its counters only add kernels config 2 does not launch (ml_spmv_kernel<4>) or are kept
# under a c4_ prefix
for tag in ("c4_trace", "batch_trace", "online_trace"):
    ks2 = sorted(glob.glob(f"{src}/{tag}/*/*kernel_stats.csv") + glob.glob(f"{src}/{tag}/*kernel_stats.csv"), key=os.path.getmtime, reverse=True)
    if ks2:
        shutil.copy(ks2[0], out + "_" + tag.replace("_trace", "") + "_kernel_stats.csv")
pmc4 = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob(f"{src}/c4_{c}/*/*counter_collection.csv"), key=os.path.getmtime, reverse=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c:
            continue
        v = float(r["Counter_Value"])
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if v < 1.0 and ("ml_" in k or "pcg_" in k or "_lm_kernel" in k):
            continue
        agg[k][0] += 1
        agg[k][1] += v
    for k, (n, v) in agg.items():
        pmc4.setdefault(k, {})[c] = dict(launches=n, mean_kb=v / max(n, 1))
for k, d in pmc4.items():
    f = d.get("FETCH_SIZE", {}).get("mean_kb", 0.0)
    w = d.get("WRITE_SIZE", {}).get("mean_kb", 0.0)
    d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    pmc["c4:" + k] = d
    if k not in pmc:
        pmc[k] = d
json.dump(pmc, open(out + "_pmc.json", "w"), indent=1, sort_keys=True)
t = {}
# (round 4: the timed solves launch the slot twins of the device-resident LM loop - ml_spmv_lm_kernel<...>, hessian_lm_kernel - the
#  profiled solve the by-value instantiations of the same bodies; a key takes the first name PREFIX that was seen)
for key, names in (("pcg_spmv_bytes_per_launch", ("uzl::ml_spmv_lm_kernel<1, 1, 8", "uzl::ml_spmv_kernel<1>", "uzl::pcg_spmv_kernel")),
                   ("pcg_spmv4_bytes_per_launch", ("c4:uzl::ml_spmv_lm_kernel<4, 4, 4", "uzl::ml_spmv_lm_kernel<4, 4, 4", "c4:uzl::ml_spmv_kernel<4>", "uzl::ml_spmv_kernel<4>")),
                   ("c4_hessian_bytes_per_launch", ("c4:uzl::hessian_kernel", "c4:uzl::hessian_lm_kernel")),      # (the slot twin also runs as a no-op ahead of the look: its mean is diluted)
                   ("hessian_bytes_per_launch", ("uzl::hessian_kernel", "uzl::hessian_lm_kernel")),
                   ("c4_ns_gemm_bytes_per_launch", ("c4:uzl::ml_ns_gemm_lm_kernel", "c4:uzl::ml_ns_gemm_kernel")),
                   ("ns_gemm32_bytes_per_launch", ("uzl::ml_ns_gemm32_lm_kernel", "uzl::ml_ns_gemm32_kernel")),
                   ("knn2_bytes_per_launch", ("uzl::knn2_mfma_kernel<8, 2", "uzl::knn2_lds_kernel<8, 1>", "uzl::knn2_kernel<8>")),
                   ("estimate_bytes_per_launch", ("uzl::estimate_kernel",)),
                   ("wire_unpack_bytes_per_launch", ("uzl::wire_unpack_kernel",))):
    for nm in names:
        hit = [k for k in sorted(pmc) if k.startswith(nm)]
        if hit:
            t[key] = pmc[hit[0]]["hbm_bytes_per_launch"]
            break
# the same kernels' average dispatch duration in the rocprofv3 --kernel-trace --stats summaries of this collection (the by-value
# instantiations the profiled solve of bench.py launches): bench.py prints them beside its own event-timed figures
def stats_avg_us(path, prefixes):
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError:
        return None
    for pre in prefixes:
        for r in rows:
            if r["Name"].replace("void ", "").startswith(pre):
                return round(float(r["AverageNs"]) / 1e3, 3)
    return None


rp = {}
for key, path, prefixes in (("pcg_spmv", out + "_kernel_stats.csv", ("uzl::ml_spmv_kernel<1>",)),
                            ("hessian", out + "_kernel_stats.csv", ("uzl::hessian_kernel",)),
                            ("ns_gemm32", out + "_kernel_stats.csv", ("uzl::ml_ns_gemm32_kernel",)),
                            ("knn2", out + "_kernel_stats.csv", ("uzl::knn2_mfma_kernel<8, 2",)),
                            ("estimate", out + "_kernel_stats.csv", ("uzl::estimate_kernel",)),
                            ("wire_unpack", out + "_kernel_stats.csv", ("uzl::wire_unpack_kernel",)),
                            ("pcg_spmv4", out + "_c4_kernel_stats.csv", ("uzl::ml_spmv_kernel<4>",)),
                            ("pcg_cg4", out + "_c4_kernel_stats.csv", ("uzl::ml_cg_kernel<4",)),
                            ("c4_hessian", out + "_c4_kernel_stats.csv", ("uzl::hessian_kernel",)),
                            ("c4_ns_gemm", out + "_c4_kernel_stats.csv", ("uzl::ml_ns_gemm_kernel",))):
    v = stats_avg_us(path, prefixes)
    if v is not None:
        rp[key + "_rocprof_avg_us"] = v
t.update(rp)
# provenance: the tag of this collection and the commit of the tree it ran on (profiles/collect.sh <tag> <commit>)
import datetime
t["_meta"] = dict(tag=os.path.basename(out), commit=(sys.argv[3] if len(sys.argv) > 3 else "unknown"),
                  collected_utc=datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
                  how="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE in separate passes, bytes = (2 FETCH + WRITE) KB x 1024 (gfx950 correction); "
                      "*_rocprof_avg_us from the --kernel-trace --stats pass of the same collection")
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(t))
