#!/usr/bin/env python3
"""Diagnostic: randomized estimator-vs-oracle sweep (bit-exact): frame sizes, descriptor widths, outlier / validity fractions,
thresholds, iteration counts, early exit, PROSAC on / off, several FeatureData per node.
  python tests/diag/stress_match.py [n_batches] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O                                   # noqa: E402
from uzliti_slam_amd import capi, synth              # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = total = 0
for b in range(n_batches):
    cfg = dict(ransac_threshold=float(rng.choice([0.03, 0.1, 0.2, 0.5])), ransac_iteration=int(rng.choice([1, 7, 64, 100, 333, 500, 1000])),
               ransac_break_percentage=float(rng.choice([0.3, 0.6, 1.0])), do_prosac=int(rng.random() < 0.7), seed=int(rng.integers(0, 2**40)))
    desc_bytes = int(rng.choice([32, 32, 64]))
    m = capi.Match(**cfg)
    pairs, ids = [], []
    for j in range(int(rng.integers(3, 20))):
        n_kp = int(rng.choice([7, 8, 9, 31, 64, 65, 200, 513, 1000, 1500, 2700]))
        f, t, _ = synth.make_pair(rng, n_kp=n_kp, desc_bytes=desc_bytes, flip_p=float(rng.choice([0.0, 0.05, 0.12])),
                                  outlier_frac=float(rng.choice([0.0, 0.4, 0.9])), invalid_frac=float(rng.choice([0.0, 0.1, 0.6])))
        if rng.random() < 0.3:                    # a smaller train set than query set
            k = max(7, n_kp // 3)
            f = dict(f, desc=f["desc"][:k].copy(), pos=f["pos"][:, :k].copy(), valid=f["valid"][:k].copy())
        if rng.random() < 0.15:                   # duplicate descriptors: ties in the 2-NN
            t["desc"][1::2] = t["desc"][0::2][: len(t["desc"][1::2])]
        fl, tl = [f], [t]
        if rng.random() < 0.3:                    # nodes with several FeatureData: other sensors / feature types, or the same twice
            f2, t2, _ = synth.make_pair(rng, n_kp=int(rng.choice([7, 40, 300])), desc_bytes=desc_bytes)
            f2["sensor_frame"] = int(rng.integers(0, 2)); t2["sensor_frame"] = int(rng.integers(0, 2))
            f2["feature_type"] = int(rng.choice([2, 3])); t2["feature_type"] = int(rng.choice([2, 3]))
            fl = [f2, f] if rng.random() < 0.5 else [f, f2]
            tl = [t, t2] if rng.random() < 0.5 else [t2, t]
        pairs.append((fl, tl))
        ids.append(([m.add_frame(x["desc"], x["pos"], x["valid"], feature_type=x["feature_type"], sensor_frame=x["sensor_frame"]) for x in fl],
                    [m.add_frame(x["desc"], x["pos"], x["valid"], feature_type=x["feature_type"], sensor_frame=x["sensor_frame"]) for x in tl]))
    job_ids = [int(x) for x in rng.integers(0, 2**50, len(pairs))]
    mc = max(len(x["desc"]) for _, tl in pairs for x in tl)
    res, diag = m.estimate(ids, job_ids=job_ids, max_corr=mc)
    for j, (fl, tl) in enumerate(pairs):
        f, t = fl[0], tl[0]
        w = O.estimate_edge(fl, tl, ransac_threshold=cfg["ransac_threshold"], ransac_iteration=cfg["ransac_iteration"],
                            break_percentage=cfg["ransac_break_percentage"], do_prosac=bool(cfg["do_prosac"]), seed=cfg["seed"], job_id=job_ids[j])
        k = w["n_corr"]
        fr_ok = (res[j]["frame_from"] < 0 and w["frame_from"] < 0) or (w["frame_from"] >= 0 and res[j]["frame_from"] == ids[j][0][w["frame_from"]] and res[j]["frame_to"] == ids[j][1][w["frame_to"]])
        ok = (fr_ok and res[j]["ok"] == w["ok"] and res[j]["consensus"] == w["consensus"] and res[j]["n_corr"] == k and res[j]["n_matches"] == w["n_matches"]
              and res[j]["iterations_run"] == w["iterations_run"] and res[j]["best_iteration"] == w["best_iteration"]
              and np.array_equal(diag["mask"][j, :k], w["mask"]) and np.array_equal(diag["corr_query"][j, :k], w["corr_query"])
              and np.array_equal(diag["corr_train"][j, :k], w["corr_train"]) and np.array_equal(res[j]["T"].reshape(3, 4), w["T"])
              and (res[j]["mse"] == w["mse"]) and np.array_equal(res[j]["information"].reshape(6, 6), w["information"]))
        total += 1; bad += 0 if ok else 1
        if not ok:
            print("MISS batch %d job %d n_to %d n_from %d cfg %s: gpu (%d, %d, %d) oracle (%d, %d, %d)" % (b, j, len(t["desc"]), len(f["desc"]), cfg,
                  res[j]["ok"], res[j]["consensus"], res[j]["n_corr"], w["ok"], w["consensus"], k), flush=True)
    m.close()
print("%d pairs, %d misses" % (total, bad))
sys.exit(1 if bad else 0)
