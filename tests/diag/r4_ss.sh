set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
UZL_SCHUR_STRONG_MIN=16 python3 tests/diag/small_strong.py 2>&1 | tail -8
