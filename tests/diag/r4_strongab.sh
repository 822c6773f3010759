set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
timeout -k 10 600 python3 tests/diag/strong_ab.py
