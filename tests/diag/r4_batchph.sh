set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_PHASES=1 python3 tests/diag/batch_phases.py 16 2>&1 | grep -E "segments over|batch of" | tail -3
