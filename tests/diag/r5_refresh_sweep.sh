#!/bin/bash
# the lazy-refresh threshold (rebuild the hierarchy while chi2 still moves by more than UZL_ML_REFRESH_REL per step) and the number of
# Newton-Schulz steps, swept on one box: config 2, 10k / 50k, a chain-like 20k graph (diagnostic build; best / median of repeated solves)
export UZL_LIB=$PWD/uzliti_slam_amd/libuzl_mi355x_diag.so
for rel in 1e-3 1e-2 3e-2 1e-1; do
  for ns in -1 2; do
    echo "== UZL_ML_REFRESH_REL=$rel UZL_ML_NS_STEPS=$ns"
    UZL_ML_REFRESH_REL=$rel UZL_ML_NS_STEPS=$ns python3 tests/diag/c2_repeat.py
  done
done
