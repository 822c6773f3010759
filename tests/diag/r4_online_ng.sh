set -e
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  UZL_NO_GRAPH=$v timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online_ng.json 2> gpurun_out/r4/online_ng.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r4/online_ng.json'))
print('UZL_NO_GRAPH=$v', {k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('wall_s','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations')})"
done
