set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for m in 0 256 384; do echo "--- UZL_SCHUR_STRONG_ONE_MAX=$m"; UZL_SCHUR_STRONG_ONE_MAX=$m NUMBERING=2 timeout -k 10 300 python3 tests/diag/online_modes.py 2>/dev/null | fold -w 220; done
