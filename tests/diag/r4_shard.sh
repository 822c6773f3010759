timeout -k 10 1100 python -m pytest tests/test_sharded_gpu.py tests/test_online_gpu.py -x -q 2>&1 | tail -15
