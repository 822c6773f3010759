set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for r in 4 2 1; do
  echo "--- UZL_SPMV4_RPW=$r"
  UZL_SPMV4_RPW=$r LOG=0 timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -E "solve:" | tail -1
  UZL_SPMV4_RPW=$r timeout -k 10 200 python3 tests/diag/c2_repeat.py 2>&1 | grep 10000 | tail -1
done
