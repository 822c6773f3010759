set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_pgo_gpu.py tests/test_lm_loops_gpu.py tests/test_sharded_gpu.py -x -q -m gpu 2>&1 | tail -4
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_PHASES=1 LOG=0 timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -E "segments over|solve:" | tail -2
UZL_ML_ASYNC_STRONG=0 UZL_PHASES=1 LOG=0 timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -E "segments over|solve:" | tail -2
timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online3.json 2> gpurun_out/r4/online3.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online3.json'))
print({k: d[k] for k in ('wall_s','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations','ate_online_m','not_converged') if k in d})"
