set -e
cd $GRAFT_REPO_ROOT
for m in 0 2; do
  echo "--- NUMBERING=$m small"; NUMBERING=$m timeout -k 10 500 python3 tests/diag/stress_pgo.py 40 7 2>&1 | grep -E "MISS|cases" | tail -5
  echo "--- NUMBERING=$m large"; NUMBERING=$m timeout -k 10 500 python3 tests/diag/stress_pgo.py 14 11 large 2>&1 | grep -E "MISS|cases" | tail -5
done
