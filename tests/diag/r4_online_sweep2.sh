set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
run() {
  env "$@" timeout -k 10 300 python3 tests/diag/online_run.py > gpurun_out/r4/online_sw.json 2> gpurun_out/r4/online_sw.err
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/r4/online_sw.json'))
print(' '.join(sys.argv[1:]), {k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('wall_s','structure_ms_per_solve','optimize_ms_per_solve','pcg_iterations','lm_iterations','not_converged') if k in d})" "$@"
}
run UZL_SCHUR_CAP=12
run UZL_SCHUR_CAP=16
run UZL_SCHUR_STRONG_THETA=15
run UZL_SCHUR_STRONG_THETA=40
run UZL_SCHUR_STRONG_MIN=32
