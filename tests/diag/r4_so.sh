set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/stream_overlap.py 10 0
python3 tests/diag/stream_overlap.py 10 -1
GPU_MAX_HW_QUEUES=2 python3 tests/diag/stream_overlap.py 6 0
