set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_lm_loops_gpu.py tests/test_append_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online5.json 2> gpurun_out/r4/online5.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online5.json'))
print({k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('wall_s','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations')}); print(d['seconds'])"
python3 tests/diag/c2_repeat.py 2>&1 | tail -3
