set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 tests/diag/batch_queue_phase.py c2 2>&1 | tail -3
bash profiles/collect.sh r04c > gpurun_out/r4/collect_r04c.log 2>&1 || { tail -20 gpurun_out/r4/collect_r04c.log; exit 1; }
head -4 gpurun_out/prof_r04c_summary/r04c_kernel_stats.csv | cut -c1-150
