set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_pgo_gpu.py tests/test_lm_loops_gpu.py tests/test_batch_gpu.py tests/test_sharded_gpu.py -x -q -m gpu 2>&1 | tail -5
python3 tests/diag/c2_repeat.py 2>&1 | tail -3
SHAPES="1000:5000" bash tests/diag/r4_trace2.sh 2>&1 | tail -14
