set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/c5_structure.py 2>&1 | tail -40
