set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for n in 100 300 500; do
  e=$((n*3)); [ $n -ge 300 ] && e=$((n*4))
  for l in 1 2; do
    echo "nodes $n edges $e lanes $l"
    UZL_BATCH_LANES=$l NODES=$n EDGES=$e python tests/diag/batch_scaling.py 16 64
  done
done
