set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_adapter_gpu.py -x -q -m gpu 2>&1 | tail -8
