set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
run() {
  echo "== $1"
  env $1 python3 tests/diag/batch_churn.py 2>&1 | grep -E "fresh|small graphs|config 2" | cut -c1-120
  env $1 python3 bench.py --no-cpu-baseline --no-c4 --no-online --no-formats --no-sharded --no-secondary --batch-queue 0 > gpurun_out/r4/bq.json 2> gpurun_out/r4/bq.err || { tail -20 gpurun_out/r4/bq.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4/bq.json').read().strip().splitlines()[-1])
b=d['batched']
print('   bench --no-secondary: c2', b['ms_per_batch'], 'small', b['small_graphs']['ms_per_batch'], 'chain', b['chain_like']['ms_per_batch'])
PY
  env $1 python3 bench.py --no-cpu-baseline --no-c4 --no-online --no-formats --no-sharded --batch-queue 0 > gpurun_out/r4/bq.json 2> gpurun_out/r4/bq.err || { tail -20 gpurun_out/r4/bq.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4/bq.json').read().strip().splitlines()[-1])
b=d['batched']
print('   bench with secondary: c2', b['ms_per_batch'], 'small', b['small_graphs']['ms_per_batch'], 'chain', b['chain_like']['ms_per_batch'])
PY
}
run "UZL_BATCH_LANES=1 UZL_BATCH_S2_PRIO=0"
run "UZL_BATCH_LANES=2 UZL_BATCH_NO_S2=1"
run "UZL_BATCH_LANES=1 UZL_BATCH_NO_S2=1"
