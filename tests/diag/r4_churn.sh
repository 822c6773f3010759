set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for l in 1 2; do
echo "== lanes $l, no torch"; UZL_BATCH_LANES=$l python3 tests/diag/batch_churn.py 2>&1 | grep -v amdgpu.ids
echo "== lanes $l, torch first"; UZL_BATCH_LANES=$l python3 tests/diag/batch_churn.py torch 2>&1 | grep -v amdgpu.ids
done
