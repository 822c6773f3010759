set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4
cd $GRAFT_REPO_ROOT
rm -rf /tmp/trb
UZL_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/trb -o b -- python3 tests/diag/batch_phases.py 16 > $OUT/btrace_run.log 2>&1
f=$(find /tmp/trb -name "*kernel_trace.csv" | head -1)
python3 tests/diag/trace_grid.py $f ml_geometry ml_galerkin ml_inverses ml_mult ml_ns_ ml_cmat32 hessian lm_ eval oplus chi2 ml_init | head -30
