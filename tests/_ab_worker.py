"""Helper of tests/test_ab_paths_gpu.py: runs one small match batch and two solves through the C ABI and prints digests.
The A/B switches (UZL_KNN2_VALU, UZL_VOTE_VALU, UZL_ML_SYNC_REBUILD, UZL_ML_NO_COMP4) are read once per process, hence a subprocess."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uzliti_slam_amd import capi, synth  # noqa: E402

out = {}
m = capi.Match(ransac_threshold=0.1, ransac_iteration=300, ransac_break_percentage=0.6, seed=11)
pairs = synth.make_pairs(12, n_kp=700, seed=3)
ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
res, diag = m.estimate(ids, max_corr=700)
h = hashlib.sha256()
for r in res:
    h.update(np.asarray(r["T"]).tobytes()); h.update(np.asarray(r["information"]).tobytes())
    h.update(np.array([r["ok"], r["consensus"], r["n_matches"], r["n_corr"], r["iterations_run"], r["best_iteration"]], np.int64).tobytes())
for k in ("corr_query", "corr_train", "corr_dist", "mask"):
    h.update(np.ascontiguousarray(diag[k]).tobytes())
out["match"] = h.hexdigest()
m.close()
for name, (n, e, its) in dict(small=(700, 3000, 8), large=(6000, 24000, 4)).items():
    g = synth.make_pose_graph(n, e, seed=n)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st = p.optimize(its)
    poses = p.store()[0]
    p.close()
    out[name] = dict(chi2=st["chi2_final"], pcg=st["pcg_iterations"], status=st["status"], poses=poses.reshape(-1).tolist())
print(json.dumps(out))
