set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/batch_lanes.py 16 1 2 4
python3 tests/diag/batch_lanes.py 32 1 2 4
