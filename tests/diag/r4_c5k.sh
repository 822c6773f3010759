set -e
cd $GRAFT_REPO_ROOT
LOG=0 timeout -k 10 200 python3 tests/diag/c5_solve_log.py 2>&1 | grep -E "solve:|schur|ml_cg|pcg_spmv|linearize" | head -8
