#!/bin/bash
# Schur elimination A/B under rocprofv3 (dispatch times): an older build kept beside the product library, chain-like graphs
set -e
R=$PWD
mkdir -p gpurun_out/r5h
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  if [ $v = new ]; then unset UZL_LIB; else export UZL_LIB=$R/uzliti_slam_amd/libuzl_ab_$v.so; fi
  rm -rf $R/gpurun_out/r5h/prof_s$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5h/prof_s$v -- python3 $R/tests/diag/kernel_times.py 20000:21700 5000:5400 1500:1530 > $R/gpurun_out/r5h/kts_$v.log 2>&1
  echo "== $v"
  find $R/gpurun_out/r5h/prof_s$v -name "*kernel_stats.csv" -exec grep -E "schur" {} + < /dev/null | cut -c1-160
  grep "^n " $R/gpurun_out/r5h/kts_$v.log
done
cd $R
python3 -m pytest tests -m gpu -x -q -k "schur or lm_loops or append or chain or sharded" > gpurun_out/r5h/tests_schur.log 2>&1 || (tail -30 gpurun_out/r5h/tests_schur.log; exit 1)
tail -2 gpurun_out/r5h/tests_schur.log
