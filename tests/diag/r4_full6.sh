set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python3 tests/diag/stress_batch.py 14 11 2>&1 | tail -2
timeout -k 10 500 python3 tests/diag/stress_pgo.py 2>&1 | tail -2
