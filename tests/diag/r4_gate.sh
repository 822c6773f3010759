set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gate_gpu.py tests/test_online_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online7.json 2> gpurun_out/r4/online7.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online7.json'))
print({k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('wall_s','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations','gate_accepted')}); print(d['seconds'])"
