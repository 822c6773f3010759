set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for SH in ${SHAPES:-1000:5000 10000:50000}; do
  rm -rf /tmp/tr
  UZL_NO_GRAPH=1 LOOPS=1 REPS=2 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o lm -- python3 tests/diag/lm_passes.py $SH > $OUT/trace2_run_$SH.log 2>&1
  f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
  python3 tests/diag/trace_grid.py $f ml_geometry ml_transform ml_reduce ml_inverses ml_mult ml_ns_ ml_cmat32 ml_dense ml_galerkin ml_pair > $OUT/trace2_$SH.txt
  cat $OUT/trace2_$SH.txt
done
