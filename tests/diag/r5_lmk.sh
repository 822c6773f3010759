#!/bin/bash
# dispatch times of the LM loop's decision kernels (rocprofv3) + the loop-equality tests
set -e
R=$PWD
mkdir -p gpurun_out/r5h
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r5h/prof_lmk
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5h/prof_lmk -- python3 $R/tests/diag/kernel_times.py 1000:5000 > $R/gpurun_out/r5h/kt_lmk.log 2>&1
find $R/gpurun_out/r5h/prof_lmk -name "*kernel_stats.csv" -exec grep -E "lm_tail|lm_head|eval_lm|ml_init_lm|hessian_lm" {} + < /dev/null | cut -c1-200
cd $R
python -m pytest tests -m gpu -x -q -k "lm_loops or batch or append or online" > gpurun_out/r5h/tests_lmk.log 2>&1 || (tail -30 gpurun_out/r5h/tests_lmk.log; exit 1)
tail -2 gpurun_out/r5h/tests_lmk.log
