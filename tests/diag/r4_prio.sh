set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for pr in -1 0; do
  echo "== handle rebuild stream priority $pr"
  UZL_S2_PRIO=$pr timeout -k 10 600 python3 tests/diag/online_run.py > gpurun_out/r4/online_p.json 2> gpurun_out/r4/online_p.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r4/online_p.json'))
print({k: d[k] for k in ('wall_s','solves','optimize_ms_per_solve','structure_ms_per_solve','pcg_iterations') if k in d})"
  UZL_S2_PRIO=$pr python3 tests/diag/c2_phases.py | tail -1
  UZL_S2_PRIO=$pr python3 tests/diag/c2_phases.py 10000 50000 | tail -1
  UZL_S2_PRIO=$pr python3 tests/diag/c2_phases.py 3000 13000 | tail -1
  UZL_S2_PRIO=$pr python3 tests/diag/small_repeat.py | tail -3
done
