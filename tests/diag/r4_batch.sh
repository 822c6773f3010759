timeout -k 10 900 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -15
NODES=1500 EDGES=1530 python tests/diag/batch_scaling.py 1 16 64
python tests/diag/batch_scaling.py 16
