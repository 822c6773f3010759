set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 tests/diag/stress_batch.py 2>&1 | tail -4
