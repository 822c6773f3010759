set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for rep in 1 2; do
for v in "UZL_STREAM_CHECK_MIN_N=1" "UZL_STREAM_CHECK_MIN_N=100000000"; do
env $v UZL_STREAM_DBG=1 timeout -k 10 600 python3 tests/diag/online_run.py > gpurun_out/r4/online_p.json 2> gpurun_out/r4/online_p.err
grep -c "independent_stream" gpurun_out/r4/online_p.err || true
python3 -c "
import json; d=json.load(open('gpurun_out/r4/online_p.json'))
print('[$v]', {k: round(d[k],3) for k in ('wall_s','optimize_ms_per_solve','structure_ms_per_solve') if k in d})"
done
done
