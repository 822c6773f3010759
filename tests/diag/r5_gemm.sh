set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python3 -m pytest tests/test_pgo_gpu.py tests/test_batch_gpu.py tests/test_lm_loops_gpu.py -m gpu -x -q > gpurun_out/r5/gemm_tests.log 2>&1 || { tail -30 gpurun_out/r5/gemm_tests.log; exit 1; }
tail -2 gpurun_out/r5/gemm_tests.log
python3 tests/diag/ns_gemm_c4.py > gpurun_out/r5/gemm_c4.log 2>&1
python3 tests/diag/ns_gemm_c4.py 20000 100000 > gpurun_out/r5/gemm_c20.log 2>&1
python3 tests/diag/ns_gemm_c4.py 1000 5000 > gpurun_out/r5/gemm_c2.log 2>&1
head -3 gpurun_out/r5/gemm_c4.log gpurun_out/r5/gemm_c20.log gpurun_out/r5/gemm_c2.log
