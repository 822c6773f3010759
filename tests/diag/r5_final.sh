cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python3 -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest_r.log 2>&1 || { tail -40 gpurun_out/r5/gputest_r.log; exit 1; }
tail -2 gpurun_out/r5/gputest_r.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_r.json 2> gpurun_out/r5/bench_r.err || { tail -20 gpurun_out/r5/bench_r.err; exit 1; }
cp gpurun_out/bench_full.json gpurun_out/r5/bench_r_full.json      # (a later bench.py run - profiles/collect.sh - overwrites gpurun_out/bench_full.json)
wc -c gpurun_out/r5/bench_r.json
python3 - <<'PY'
import json
c=json.loads(open('gpurun_out/r5/bench_r.json').read().strip().splitlines()[-1])
print('C2', c['value'], c['ms_per_step'], 'first', c['first_solve_ms'], 'repeat', c['repeat_identical'], 'lm_overhead', c['lm_overhead_ms'], c['roofline'])
print('rooflines', c['rooflines'])
print('C4', {k:v for k,v in c['c4_1gpu'].items() if k not in ('roofline',)})
print('batched', c['batched'])
print('C5', c['online_c5'])
print('secondary', c['secondary'])
print('streams', c['streams'], 'xy', c['xy_only'], 'formats', c['formats'])
PY
