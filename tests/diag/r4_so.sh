set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/stream_overlap.py 8 0 chain 60 1000
python3 tests/diag/stream_overlap.py 8 200 chain 60 1000
python3 tests/diag/stream_overlap.py 8 0 chain 60 100
