set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -3
python3 tests/diag/batch_phases.py 16 2>&1 | tail -1
