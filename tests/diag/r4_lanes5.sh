set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for l in 1 2; do
echo "== lanes $l"
UZL_BATCH_LANES=$l python3 tests/diag/batch_queue_phase.py chain
UZL_BATCH_LANES=$l python3 tests/diag/batch_queue_phase.py c2
done
