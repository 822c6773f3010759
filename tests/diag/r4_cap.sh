set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for c in 24 12 8; do echo "--- UZL_SCHUR_CAP=$c"; UZL_SCHUR_CAP=$c python3 tests/diag/small_strong.py 2>&1 | grep "numbering 2"; UZL_SCHUR_CAP=$c NUMBERING=0 timeout -k 10 300 python3 tests/diag/online_modes.py 2>/dev/null | tail -1; done
