set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_append_gpu.py tests/test_pgo_gpu.py -x -q -m gpu -k "append or reduced_numbering or chain_like" 2>&1 | tail -4
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
timeout -k 10 600 python3 tests/diag/strong_ab.py 2>&1 | head -7
unset UZL_LIB
timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online6.json 2> gpurun_out/r4/online6.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online6.json'))
print({k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('wall_s','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations')})"
grep -c "solve" gpurun_out/r4/online6.err
