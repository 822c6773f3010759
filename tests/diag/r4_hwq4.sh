set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for sec in "--no-secondary" ""; do
for hp in -1 0; do
for bp in -1 0; do
UZL_S2_PRIO=$hp UZL_BATCH_S2_PRIO=$bp python3 bench.py --steps 20 --no-cpu-baseline --no-c4 --no-online --no-formats --no-sharded $sec --batch-queue 0 > gpurun_out/r4/bq.json 2> gpurun_out/r4/bq.err || { tail -20 gpurun_out/r4/bq.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4/bq.json').read().strip().splitlines()[-1])
b=d['batched']
print('[$sec] handle s2 priority $hp, batch s2 priority $bp: c2', b['ms_per_batch'], 'small', b['small_graphs']['ms_per_batch'], 'chain', b['chain_like']['ms_per_batch'], 'primary', d['ms_per_step'])
PY
done
done
done
