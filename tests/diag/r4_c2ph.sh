set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/c2_phases.py 2>&1 | tail -1
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_PHASES=1 python3 tests/diag/c2_phases.py 2>&1 | grep -E "segments over|best" | tail -3
UZL_PHASES=1 UZL_PHASES_EACH=1 python3 tests/diag/c2_phases.py 2>&1 | grep -B40 "segments over" | tail -45 | cut -c1-200
