import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
from uzliti_slam_amd import capi, synth
for n, e in ((1000, 5000), (2000, 9000), (300, 1200)):
    g = synth.make_pose_graph(n, e, seed=12345)
    p = capi.Pgo(pass_history=1)
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20)
    ts = []
    for _ in range(12):
        p.reset(); t0 = time.perf_counter(); st = p.optimize(20); ts.append(time.perf_counter() - t0)
    h = hashlib.sha256(p.store()[0].tobytes()).hexdigest()[:12]
    print("%d/%d: best %.3f ms median %.3f ms, pcg %d, poses %s" % (n, e, 1e3 * min(ts), 1e3 * np.median(ts), st["pcg_iterations"], h))
    p.close()
