python - <<'PY'
import time, numpy as np
from uzliti_slam_amd import capi, synth
pairs = synth.make_pairs(512, n_kp=1000, desc_bytes=32, seed=777)
m = capi.Match(ransac_threshold=0.1, ransac_iteration=500, ransac_break_percentage=1.0, do_prosac=1, seed=777)
ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
m.estimate(ids)
m.set_profiling(True)
ks=[]
for _ in range(8):
    m.estimate(ids); ks.append(m.kernel_times()["knn2"]["ms"])
print("knn2 ms:", ["%.4f" % k for k in ks])
PY
