set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/small_repeat.py 2>&1 | grep "edges/s"
