import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["UZL_VERBOSE"] = "1"
from uzliti_slam_amd import capi, synth
B = 16
bt = capi.PgoBatch(B)
for k in range(B):
    g = synth.make_pose_graph(1000, 5000, seed=100 + 7 * k, outlier_frac=0.05 + 0.02 * (k % 3))
    bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
st = bt.optimize(20)
print("n_batched", bt.n_batched, [s["pcg_iterations"] for s in st])
