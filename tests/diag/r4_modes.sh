set -e
cd $GRAFT_REPO_ROOT
for m in ${MODES:-0 2 1}; do echo "--- reduced_numbering=$m"; NUMBERING=$m timeout -k 10 300 python3 tests/diag/online_modes.py 2>/dev/null | tail -1; done
