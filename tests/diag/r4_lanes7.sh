set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for l in 1 2; do
UZL_BATCH_LANES=$l python3 bench.py --no-cpu-baseline --no-c4 --no-online --no-formats --no-sharded --batch-queue 0 > gpurun_out/r4/bl_$l.json 2> gpurun_out/r4/bl_$l.err || { tail -20 gpurun_out/r4/bl_$l.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4/bl_$l.json').read().strip().splitlines()[-1])
b=d['batched']
print('lanes $l:', b['value'], b['ms_per_batch'], 'small', b.get('small_graphs',{}).get('value'), 'chain', b.get('chain_like',{}).get('value'))
PY
done
for l in 1 2; do
UZL_BATCH_LANES=$l python3 bench.py --no-cpu-baseline --no-c4 --no-online --no-formats --no-sharded --no-secondary --batch-queue 0 > gpurun_out/r4/bm_$l.json 2> gpurun_out/r4/bm_$l.err || { tail -20 gpurun_out/r4/bm_$l.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4/bm_$l.json').read().strip().splitlines()[-1])
b=d['batched']
print('no secondary, lanes $l:', b['value'], b['ms_per_batch'], 'small', b.get('small_graphs',{}).get('value'), 'chain', b.get('chain_like',{}).get('value'))
PY
done
