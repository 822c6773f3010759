#!/bin/bash
# Hessian build A/B under rocprofv3 (dispatch times): UZL_LIB = the library to load (older builds kept beside the product one).
set -e
R=$PWD
mkdir -p gpurun_out/r5h
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-old w1 new}; do
  if [ $v = new ]; then unset UZL_LIB; else export UZL_LIB=$R/uzliti_slam_amd/libuzl_ab_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5h/prof_$v -- python3 $R/tests/diag/kernel_times.py 1000:5000 10000:50000 20000:21700 > $R/gpurun_out/r5h/kt_$v.log 2>&1
  echo "== $v"
  find $R/gpurun_out/r5h/prof_$v -name "*kernel_stats.csv" -exec grep -E "hessian|slot_records" {} + < /dev/null | cut -c1-200
  grep "^n " $R/gpurun_out/r5h/kt_$v.log
done
