import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["UZL_VERBOSE"] = "1"
from uzliti_slam_amd import capi, synth
for k in range(16):
    g = synth.make_pose_graph(1000, 5000, seed=100 + 7 * k, outlier_frac=0.05 + 0.02 * (k % 3))
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    print("=== graph", k, file=sys.stderr, flush=True)
    st = p.optimize(20)
    print("graph", k, st["pcg_iterations"], st["lm_trials"], st["precond_builds"], st["pcg_not_converged"], flush=True)
    p.close()
