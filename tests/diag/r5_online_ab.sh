#!/bin/bash
# A/B of one diagnostic switch on the online run (config 5): wall clock and optimize seconds, settings alternated on one box
#   bash tests/diag/r5_online_ab.sh "UZL_SPMV4_RPW=4" "X=0" ...
export UZL_LIB=$PWD/uzliti_slam_amd/libuzl_mi355x_diag.so
for rep in 1 2; do
  for kv in "$@"; do
    env $kv python3 tests/diag/online_run.py 2>/dev/null | python3 -c "
import sys, json
s = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s wall %.3f s  optimize %.3f s  pcg %d' % ('$kv', s['wall_s'], s['seconds']['optimize'], s['pcg_iterations']))"
  done
done
