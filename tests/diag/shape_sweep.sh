# solver against the oracle's direct solve over graph shapes (dense loops ... chain-like); A/B switches through the environment
for g in "1000 5000 20" "8000 8400 10" "3000 3100 10" "2000 2040 10" "5000 25000 10" "1500 1530 20" "20000 21700 6"; do
  timeout -k 5 300 python tests/diag/sparse_loops.py $g 2>&1 | tail -1
done
NO_ORACLE=1 timeout -k 5 200 python tests/diag/sparse_loops.py 10000 50000 20 2>&1 | tail -1
