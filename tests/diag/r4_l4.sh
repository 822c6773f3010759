set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
echo "== four sequences"
UZL_BATCH_LANES=4 UZL_BATCH_LANE_MIN=8 UZL_STREAM_DBG=1 python3 tests/diag/batch_phases.py 16 2>&1 | grep -E "kept|batch of" | tail -8 | cut -c1-150
UZL_BATCH_LANES=4 UZL_BATCH_LANE_MIN=8 python3 tests/diag/batch_churn.py 2>&1 | grep -E "fresh|small graphs|config 2" | cut -c1-140
UZL_BATCH_LANES=4 UZL_BATCH_LANE_MIN=8 python3 tests/diag/batch_phases.py 32 2>&1 | tail -1
echo "== two"
UZL_BATCH_LANES=2 python3 tests/diag/batch_churn.py 2>&1 | grep -E "fresh|small graphs|config 2" | cut -c1-140
UZL_BATCH_LANES=2 python3 tests/diag/batch_phases.py 32 2>&1 | tail -1
UZL_BATCH_LANES=4 UZL_BATCH_LANE_MIN=8 timeout -k 10 300 python3 tests/diag/stress_batch.py 8 11 2>&1 | tail -2
