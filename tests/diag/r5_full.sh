set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python3 -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest.log 2>&1 || { tail -40 gpurun_out/r5/gputest.log; exit 1; }
tail -3 gpurun_out/r5/gputest.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err || { tail -20 gpurun_out/r5/bench.err; exit 1; }
wc -c gpurun_out/r5/bench.json
cat gpurun_out/r5/bench.json
