set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
echo "--- row order"; UZL_SCHUR_STRONG_MIN=0 LOG=0 timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -v "structure:" | tail -16
echo "--- strong aggregates"; LOG=${LOG:-0} timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -v "structure:" | tail -${TAILN:-16}
