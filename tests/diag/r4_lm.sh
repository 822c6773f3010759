set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_lm_loops_gpu.py -x -q > gpurun_out/r4/lm_loops.log 2>&1 || { tail -40 gpurun_out/r4/lm_loops.log; exit 1; }
tail -5 gpurun_out/r4/lm_loops.log
python tests/diag/c2_repeat.py > gpurun_out/r4/lm_c2_repeat.log 2>&1
python tests/diag/lm_overhead.py 100:300 1000:5000 10000:50000 >> gpurun_out/r4/lm_c2_repeat.log 2>&1
cat gpurun_out/r4/lm_c2_repeat.log
