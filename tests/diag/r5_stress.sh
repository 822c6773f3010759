#!/bin/bash
# randomized sweeps against the oracle + the reproducibility harness, on the code as it stands: gpurun_out/r5/stress_f.log, repro_f.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
( python3 tests/diag/stress_pgo.py 60 11 && python3 tests/diag/stress_pgo.py 12 12 large && python3 tests/diag/stress_match.py ) > gpurun_out/r5/stress_f.log 2>&1
echo "stress rc $?"; grep -E "misses" gpurun_out/r5/stress_f.log
bash tests/diag/r5_repro.sh > gpurun_out/r5/repro_f.log 2>&1
echo "repro rc $?"; tail -12 gpurun_out/r5/repro_f.log
