set -e
cd $GRAFT_REPO_ROOT
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so
for u in 2 4 3; do
  echo "--- UZL_KNN2_UT=$u"
  UZL_KNN2_UT=$u timeout -k 10 300 python3 bench.py --no-c4 --no-online --no-batched --no-cpu-baseline --no-formats --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']
print({k: s[k] for k in ('value','ms_per_step','ok_fraction')}, [ (r['kernel'], r.get('avg_launch_us'), r.get('frac')) for r in d['rooflines'] if 'knn2' in r['kernel'] or 'estimate' in r['kernel']])"
done
UZL_KNN2_UT=4 timeout -k 10 600 python3 -m pytest tests/test_match_gpu.py -x -q -m gpu 2>&1 | tail -2
