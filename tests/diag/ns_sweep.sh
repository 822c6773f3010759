# Newton-Schulz step count of the composite operator (diagnostic library) against solve time, by size class
export UZL_LIB=$PWD/uzliti_slam_amd/libuzl_mi355x_diag.so
for ns in 2 4 6; do
  echo "== NS $ns"
  for g in "5000 25000 20" "10000 50000 20" "20000 100000 20" "20000 21700 20"; do
    NO_ORACLE=1 UZL_ML_NS_STEPS=$ns timeout -k 5 300 python tests/diag/sparse_loops.py $g 2>&1 | tail -1
  done
done
