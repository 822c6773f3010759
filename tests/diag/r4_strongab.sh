set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 tests/diag/strong_ab.py
