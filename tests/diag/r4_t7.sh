set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -2
python3 tests/diag/batch_queue_phase.py c2 2>&1 | tail -3
