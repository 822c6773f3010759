# stress_batch.py 0 misses; a batch of 16 within 3 % across five fresh processes (with / without other streams made first)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python3 -m pytest tests/test_gate_gpu.py tests/test_lm_loops_gpu.py tests/test_zz_streams_gpu.py -m gpu -x -q 2>&1 | tail -3
python3 tests/diag/stress_batch.py 12 5 2>&1 | tail -3
for i in 1 2 3 4 5; do
  if [ $i -ge 4 ]; then EXTRA="torch"; else EXTRA=""; fi
  python3 tests/diag/batch_churn.py $EXTRA 2>&1 | grep -E "fresh|config 2" | cut -c1-120
done
