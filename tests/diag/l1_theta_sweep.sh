for th in 0 0.55 1.0; do
  echo "=== theta $th"
  for g in "1000 5000 20" "8000 8400 10" "3000 3100 10" "2000 2040 10" "5000 25000 10"; do
    UZL_ML_L1_THETA=$th timeout -k 5 200 python tests/diag/sparse_loops.py $g 2>&1 | tail -1
  done
  NO_ORACLE=1 UZL_ML_L1_THETA=$th timeout -k 5 200 python tests/diag/sparse_loops.py 10000 50000 20 2>&1 | tail -1
done
echo "=== verbose theta 0 (guard values)"; UZL_ML_L1_THETA=0 UZL_VERBOSE=1 timeout -k 5 100 python tests/diag/sparse_loops.py 8000 8400 4 2>&1 | tail -8
echo "=== verbose theta 0.55 C2"; UZL_VERBOSE=1 timeout -k 5 100 python tests/diag/sparse_loops.py 1000 5000 6 2>&1 | tail -8
