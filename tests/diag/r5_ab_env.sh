#!/bin/bash
# A/B of one diagnostic-build switch on one box: tests/diag/c2_repeat.py (config 2 and 10k / 50k, best / median) with the switch at 0, 1, 0, 1.
#   bash tests/diag/r5_ab_env.sh UZL_LM_ODD_K
set -e
export UZL_LIB=$PWD/uzliti_slam_amd/libuzl_mi355x_diag.so
for v in 0 1 0 1; do
  echo "== $1=$v"
  env $1=$v python3 tests/diag/c2_repeat.py
done
