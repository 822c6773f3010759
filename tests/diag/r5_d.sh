cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python3 -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest_d.log 2>&1 || { tail -40 gpurun_out/r5/gputest_d.log; exit 1; }
tail -2 gpurun_out/r5/gputest_d.log
python3 tests/diag/create_cost.py 2>&1 | tail -8
for sz in "1000 5000" "10000 50000"; do UZL_VERBOSE=1 python3 tests/diag/structure_ticks.py $sz 2>&1 | grep -E "structure:|diag\]" | tail -6; done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_d.json 2> gpurun_out/r5/bench_d.err || { tail -20 gpurun_out/r5/bench_d.err; exit 1; }
python3 - <<'PY'
import json
c=json.loads(open('gpurun_out/r5/bench_d.json').read().strip().splitlines()[-1])
print('C2', c['ms_per_step'], 'first', c['first_solve_ms'], 'repeat', c['repeat_identical'], 'lm_overhead', c['lm_overhead_ms'])
print('C4', c['c4_1gpu']['ms_per_solve'], 'first', c['c4_1gpu']['first_solve_ms'], [ (r['kernel'], r['frac'], r['avg_launch_us']) for r in c['c4_1gpu']['rooflines']])
print('batched', c['batched'])
print('C5', c['online_c5'])
print('secondary', c['secondary']['value'], c['secondary']['deployed'])
PY
