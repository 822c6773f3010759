set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -5
python3 tests/diag/batch_phases.py 16 2>&1 | grep -E "batch of" | tail -1
python3 tests/diag/batch_phases.py 12 2>&1 | grep -E "batch of" | tail -1
python3 tests/diag/batch_phases.py 32 2>&1 | grep -E "batch of" | tail -1
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_BATCH_LANES=1 python3 tests/diag/batch_phases.py 16 2>&1 | grep -E "batch of" | tail -1
UZL_BATCH_LANES=2 python3 tests/diag/batch_phases.py 16 2>&1 | grep -E "batch of" | tail -1
