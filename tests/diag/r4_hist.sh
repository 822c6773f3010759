set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/c2_phases.py | tail -1
python3 tests/diag/c2_phases.py 10000 50000 | tail -1
python3 tests/diag/small_repeat.py | tail -3
timeout -k 10 600 python3 tests/diag/online_run.py > gpurun_out/r4/online_p.json 2> gpurun_out/r4/online_p.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online_p.json'))
print({k: d[k] for k in ('wall_s','solves','optimize_ms_per_solve','structure_ms_per_solve','pcg_iterations') if k in d})"
timeout -k 10 600 python -m pytest tests/test_lm_loops_gpu.py tests/test_pgo_gpu.py -x -q 2>&1 | tail -3
