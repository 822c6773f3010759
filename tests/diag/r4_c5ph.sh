set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_PHASES=1 LOG=0 timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -E "segments over|solve:" | tail -3
UZL_VERBOSE=0 LOG=1 timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -E "structure:|device-resident" | tail -12
