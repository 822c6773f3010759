set -e
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 3 --warmup 1 --rehearse-gloo --no-cpu-baseline --online-cpu-seconds 2 > gpurun_out/r4/bench_n2.json 2> gpurun_out/r4/bench_n2.err || { tail -30 gpurun_out/r4/bench_n2.err; exit 1; }
tail -3 gpurun_out/r4/bench_n2.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_n2.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','n_gpus','ms_per_step','scaling')})
print('sharded_c4' in d, {k: v for k, v in d.get('sharded_c4', {}).items() if not isinstance(v, (dict, list))} if 'sharded_c4' in d else None)
print('online', d.get('online_c5', {}).get('wall_s'), d.get('online_c5', {}).get('parallelism'))
PY
