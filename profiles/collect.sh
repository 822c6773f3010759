#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline block.  Run on the GPU box from the repo root:
#   bash profiles/collect.sh <tag> [commit]  (writes gpurun_out/prof_<tag>_summary/: kernel stats, counters, traffic.json;
#                                             `commit` = git rev-parse --short HEAD of the tree sent to the box, recorded in traffic.json)
# hipGraph replay crashes rocprofv3's kernel tracer on this image, so profiling runs use eager launches
# (UZL_NO_GRAPH=1); counters are collected in their own passes (no --kernel-trace mixed with --pmc).
set -u
TAG=${1:-r01}
COMMIT=${2:-unknown}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp UZL_NO_GRAPH=1
# config 2 (primary) + config 3 (secondary) + formats; the 10k/50k block, the batched block and the online block have their own runs below
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-c4 --no-online --no-batched"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OLDPWD/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-c4 --no-online --no-batched > $OUT/$c.json 2> $OUT/$c.err
done
# 10k / 50k on one GPU: kernel stats + the same two counters (linearize and ml_spmv_kernel<4> of this size go into traffic.json
# under their own keys only when config 2 did not provide them)
C4="--nodes 10000 --edges 50000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-c4 --no-online --no-batched"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -- python3 $OLDPWD/bench.py $C4 > $OUT/c4_trace.json 2> $OUT/c4_trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/c4_$c -- python3 $OLDPWD/bench.py --nodes 10000 --edges 50000 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-c4 --no-online --no-batched > $OUT/c4_$c.json 2> $OUT/c4_$c.err
done
# the batched block (16 config-2 graphs) and the online block (config 5), kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/batch_trace -- python3 $OLDPWD/tests/diag/batch_scaling.py 16 > $OUT/batch_trace.log 2> $OUT/batch_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/online_trace -- python3 $OLDPWD/tests/diag/online_run.py > $OUT/online_trace.json 2> $OUT/online_trace.err
cd $OLDPWD
# summaries only travel back (gpurun merges at most 64 MiB of gpurun_out/): the raw traces are deleted here; copy the files of
# gpurun_out/prof_<tag>_summary/ into profiles/ afterwards
SUM=$PWD/gpurun_out/prof_${TAG}_summary
rm -rf $SUM; mkdir -p $SUM
python3 profiles/summarize.py $OUT $SUM/$TAG $COMMIT
cp profiles/traffic.json $SUM/traffic.json
for f in trace c4_trace online_trace; do cp $OUT/$f.json $SUM/${TAG}_${f}_bench_under_rocprof.json 2>/dev/null; done
cp $OUT/batch_trace.log $SUM/${TAG}_batch_under_rocprof.log 2>/dev/null
rm -rf $OUT
