set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_STREAM_DBG=1 UZL_BATCH_LANES=2 python3 tests/diag/batch_queue_phase.py c2 2>&1 | grep -v "batch streams" | head -8
UZL_BATCH_LANES=1 python3 tests/diag/batch_queue_phase.py c2 2>&1 | head -2
UZL_BATCH_LANES=2 UZL_BATCH_S2_PRIO=-1 python3 tests/diag/batch_queue_phase.py c2 2>&1 | head -2
python3 tests/diag/batch_phases.py 16 2>&1 | tail -1
