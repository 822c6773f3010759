set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_batch_gpu.py tests/test_lm_loops_gpu.py -x -q 2>&1 | tail -3
python3 tests/diag/batch_phases.py 16 2>&1 | tail -1
python3 tests/diag/batch_churn.py 2>&1 | grep -E "fresh|small graphs|config 2" | cut -c1-120
python3 tests/diag/c2_phases.py | tail -1
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_LM_NO_HISTORY=1 python3 tests/diag/batch_phases.py 16 2>&1 | tail -1
UZL_LM_NO_HISTORY=1 python3 tests/diag/batch_churn.py 2>&1 | grep -E "fresh|small graphs|config 2" | cut -c1-120
