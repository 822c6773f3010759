cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python3 tests/diag/ns_gemm_c4.py 2>&1 | head -1
python3 tests/diag/ns_gemm_c4.py 20000 100000 2>&1 | head -1
python3 tests/diag/create_cost.py 2>&1 | tail -14
for sz in "1000 5000" "10000 50000" "8000 8400"; do UZL_VERBOSE=1 python3 tests/diag/structure_ticks.py $sz 2>&1 | grep -E "structure:|diag\]" | tail -12; done
