set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_batch_gpu.py tests/test_adapter_gpu.py -x -q 2>&1 | tail -5
