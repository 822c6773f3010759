timeout -k 10 1100 python -m pytest tests/test_pgo_gpu.py tests/test_lm_loops_gpu.py tests/test_batch_gpu.py tests/test_schur_gpu.py -x -q 2>&1 | tail -8
python tests/diag/c2_repeat.py
python tests/diag/lm_passes.py 600:2600 900:4000 1000:5000 2>&1 | grep "lm_loop=0"
python tests/diag/batch_scaling.py 16
