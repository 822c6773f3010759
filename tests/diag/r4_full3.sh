set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
python3 tests/diag/batch_queue_phase.py c2 2>&1 | tail -8
python3 tests/diag/stream_overlap.py 10 0
