set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/stream_overlap.py 8 200
python3 tests/diag/batch_churn.py 2>&1 | grep -E "fresh|small graphs|config 2" | cut -c1-120
python3 tests/diag/batch_queue_phase.py c2 2>&1 | tail -8
for sec in "--no-secondary" ""; do
python3 bench.py --no-cpu-baseline --no-c4 --no-online --no-formats --no-sharded $sec --batch-queue 0 > gpurun_out/r4/bq.json 2> gpurun_out/r4/bq.err || { tail -20 gpurun_out/r4/bq.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4/bq.json').read().strip().splitlines()[-1])
b=d['batched']
print('bench [$sec]: c2', b['ms_per_batch'], 'small', b['small_graphs']['ms_per_batch'], 'chain', b['chain_like']['ms_per_batch'], 'primary', d['ms_per_step'])
PY
done
