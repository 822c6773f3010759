set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_PHASES=1 python3 tests/diag/small_repeat.py 2>&1 | grep -E "segments over|edges/s" | awk '/edges\/s/ {print last; print} {last=$0}'
