set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 bench.py > gpurun_out/r4/bench_c.json 2> gpurun_out/r4/bench_c.err || { tail -30 gpurun_out/r4/bench_c.err; exit 1; }
tail -5 gpurun_out/r4/bench_c.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_c.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','unit','ms_per_step','lm_overhead_ms')})
print('roofline', d['roofline'])
print('parity', d.get('parity'))
print('batched', {k: v for k, v in d['batched'].items() if not isinstance(v, (dict, list))})
print('c4', {k: v for k, v in d['c4_1gpu'].items() if not isinstance(v, (dict, list))})
o=d['online_c5']; print('c5', {k: o[k] for k in ('wall_s','solves','add_graph_ms_per_solve','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations')}); print(o['seconds']); print(o['cpu_baseline'].get('pose_difference_at_that_point'))
print('secondary', {k: v for k, v in d['secondary'].items() if not isinstance(v, (dict, list))})
PY
