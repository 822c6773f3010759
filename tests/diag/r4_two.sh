set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests/test_pgo_gpu.py tests/test_lm_loops_gpu.py tests/test_batch_gpu.py tests/test_sharded_gpu.py tests/test_online_gpu.py tests/test_append_gpu.py -x -q -m gpu 2>&1 | tail -6
python3 tests/diag/small_repeat.py 2>&1 | grep "edges/s"
MODES="0" bash tests/diag/r4_modes.sh 2>&1 | tail -2
