set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_pgo_gpu.py tests/test_online_gpu.py tests/test_append_gpu.py -x -q -m gpu 2>&1 | tail -4
