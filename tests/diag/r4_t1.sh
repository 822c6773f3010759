set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_pgo_gpu.py -x -q -m gpu -k "hundreds or multi_edges" 2>&1 | tail -4
