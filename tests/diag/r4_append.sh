set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_append_gpu.py tests/test_online_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online4.json 2> gpurun_out/r4/online4.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online4.json'))
print({k: d[k] for k in ('wall_s','add_graph_ms_per_solve','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations','ate_online_m','not_converged','gate_accepted','feature_edges_valid') if k in d})
print(d['seconds'])"
