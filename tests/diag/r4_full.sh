set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15
