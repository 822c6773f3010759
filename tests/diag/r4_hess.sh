timeout -k 10 1100 python -m pytest tests/test_pgo_gpu.py tests/test_lm_loops_gpu.py tests/test_batch_gpu.py tests/test_schur_gpu.py tests/test_sharded_gpu.py -x -q 2>&1 | tail -12
python tests/diag/c2_repeat.py
python tests/diag/batch_scaling.py 16
python - <<'PY'
from uzliti_slam_amd import capi, synth
for n, e in ((1000, 5000), (10000, 50000), (20000, 21800)):
    g = synth.make_pose_graph(n, e)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    p.optimize(20); p.reset()
    p.set_profiling(True)
    p.optimize(20)
    kt = p.kernel_times()
    print(n, e, {k: "%.1f us x %d" % (1e3 * v["ms"] / v["launches"], v["launches"]) for k, v in kt.items() if k in ("linearize", "assemble", "chi2", "oplus", "finalize")})
    p.close()
PY
