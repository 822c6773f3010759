set -e
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r04c > gpurun_out/r4/collect_r04c.log 2>&1 || { tail -20 gpurun_out/r4/collect_r04c.log; exit 1; }
ls -la gpurun_out/prof_r04c_summary/
