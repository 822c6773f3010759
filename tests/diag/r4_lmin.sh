set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for m in 2 100; do
echo "== lane min $m"
for B in 4 8; do UZL_BATCH_LANE_MIN=$m python3 tests/diag/batch_scaling.py $B | tail -1; done
for B in 4 8; do UZL_BATCH_LANE_MIN=$m NODES=1500 EDGES=1530 python3 tests/diag/batch_scaling.py $B | tail -1; done
for B in 8; do UZL_BATCH_LANE_MIN=$m NODES=300 EDGES=1200 python3 tests/diag/batch_scaling.py $B | tail -1; done
for B in 8; do UZL_BATCH_LANE_MIN=$m NODES=100 EDGES=300 python3 tests/diag/batch_scaling.py $B | tail -1; done
done
