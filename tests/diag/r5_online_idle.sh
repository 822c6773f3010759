#!/bin/bash
# where the GPU idles during the online run (config 5): kernel trace of tests/diag/online_run.py, gaps by the kernel they precede
set -e
R=$PWD
mkdir -p gpurun_out/r5h
cd /tmp && export TMPDIR=/tmp UZL_NO_GRAPH=1
rm -rf $R/gpurun_out/r5h/prof_online
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5h/prof_online -- python3 $R/tests/diag/online_run.py > $R/gpurun_out/r5h/online_run.json 2> $R/gpurun_out/r5h/online_run.err
f=$(find $R/gpurun_out/r5h/prof_online -name "*kernel_trace.csv" | head -1)
python3 $R/tests/diag/trace_idle.py $f 0.5
rm -rf $R/gpurun_out/r5h/prof_online
