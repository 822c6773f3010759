set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/stream_overlap.py 10 200
python3 tests/diag/stream_overlap.py 10 201
