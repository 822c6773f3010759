set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 tests/diag/stress_batch.py 14 11 2>&1 | tail -4
