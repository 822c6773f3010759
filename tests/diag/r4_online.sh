set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tests/diag/online_run.py > gpurun_out/r4/online2.json 2> gpurun_out/r4/online2.err
tail -12 gpurun_out/r4/online2.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online2.json'))
print({k: d[k] for k in ('wall_s','solves','add_graph_ms_per_solve','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations','ate_online_m','gate_accepted','feature_edges_valid') if k in d})
print(d['seconds'])"
