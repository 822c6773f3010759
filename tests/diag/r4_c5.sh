python tests/diag/c2_repeat.py
python tests/diag/lm_passes.py 100:300 3000:3300 20000:21800 2000:9000 2>&1 | grep "lm_loop=0"
python tests/diag/c5_tolerance.py
