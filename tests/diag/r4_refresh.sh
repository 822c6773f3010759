set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for v in 1e-3 1e-4 3e-5 1e-3 1e-4; do
  echo "--- UZL_ML_REFRESH_REL=$v"
  UZL_ML_REFRESH_REL=$v NUMBERING=0 timeout -k 10 300 python3 tests/diag/online_modes.py 2>/dev/null | tail -1
  UZL_ML_REFRESH_REL=$v python3 tests/diag/c2_repeat.py 2>&1 | tail -2
done
