set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -5
python3 tests/diag/stream_overlap.py 10 200
python3 tests/diag/batch_queue_phase.py c2 2>&1 | tail -8
python3 tests/diag/batch_queue_phase.py chain 2>&1 | tail -8
