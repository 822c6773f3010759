set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/batch_lanes.py 8 1 2
python3 tests/diag/batch_lanes.py 10 1 2
python3 tests/diag/batch_lanes.py 4 1 2
NODES=1500 EDGES=1530 python tests/diag/batch_scaling.py 16
NODES=100 EDGES=300 python tests/diag/batch_scaling.py 64
