set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
run() { echo "--- $*"; env "$@" NUMBERING=2 timeout -k 10 300 python3 tests/diag/online_modes.py 2>/dev/null | tail -1; }
run UZL_SCHUR_STRONG_ONE_MAX=256
run UZL_SCHUR_STRONG_ONE_MAX=384 UZL_ML_NS_STEPS=1
run UZL_SCHUR_STRONG_ONE_MAX=256 UZL_ML_NS_STEPS=1
run UZL_SCHUR_STRONG_ONE_MAX=320
