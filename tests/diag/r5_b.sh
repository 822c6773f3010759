set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python3 -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest_b.log 2>&1 || { tail -40 gpurun_out/r5/gputest_b.log; exit 1; }
tail -2 gpurun_out/r5/gputest_b.log
python3 tests/diag/ns_gemm_c4.py > gpurun_out/r5/gemm_c4.log 2>&1
python3 tests/diag/ns_gemm_c4.py 20000 100000 > gpurun_out/r5/gemm_c20.log 2>&1
head -1 gpurun_out/r5/gemm_c4.log gpurun_out/r5/gemm_c20.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_b.json 2> gpurun_out/r5/bench_b.err || { tail -20 gpurun_out/r5/bench_b.err; exit 1; }
cat gpurun_out/r5/bench_b.json
