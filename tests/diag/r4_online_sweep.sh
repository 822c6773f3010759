set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for m in ${SWEEP:-128 256 512}; do
  UZL_SCHUR_STRONG_MIN=$m timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online_s$m.json 2> gpurun_out/r4/online_s$m.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r4/online_s$m.json'))
print('strong_min $m', {k: d[k] for k in ('wall_s','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations','ate_online_m','not_converged') if k in d})"
done
