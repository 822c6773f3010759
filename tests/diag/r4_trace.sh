set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
UZL_NO_GRAPH=1 LOOPS=${LOOPS:-0} REPS=2 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o lm -- python3 tests/diag/lm_passes.py ${SHAPE:-1000:5000} > $OUT/trace_run.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 tests/diag/trace_lm.py $f > $OUT/trace_lm_${LOOPS:-0}.txt
cat $OUT/trace_lm_${LOOPS:-0}.txt
