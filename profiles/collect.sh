#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline block.  Run on the GPU box from the repo root:
#   bash profiles/collect.sh <tag>          (writes gpurun_out/prof_<tag>/; summarise with profiles/summarize.py)
# hipGraph replay crashes rocprofv3's kernel tracer on this image, so profiling runs use eager launches
# (UZL_NO_GRAPH=1); counters are collected in their own passes (no --kernel-trace mixed with --pmc).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp UZL_NO_GRAPH=1
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OLDPWD/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$c.json 2> $OUT/$c.err
done
cd $OLDPWD
python3 profiles/summarize.py $OUT profiles/$TAG
