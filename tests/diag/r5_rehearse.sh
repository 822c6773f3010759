# bench.py --gpus 2 on a one-GPU box: process group over gloo, both ranks on device 0 (the numbers mean nothing; the N > 1 control flow runs).
# Second run: the sharded block's child job forced on - on one GPU its two ranks cannot form a communicator, the error must come back as a
# record and both ranks must meet again behind it.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --rehearse-gloo --batch-queue 0 > gpurun_out/r5/rehearse2.json 2> gpurun_out/r5/rehearse2.err
echo "rc $?"; tail -c 600 gpurun_out/r5/rehearse2.json; echo
UZL_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse-gloo --no-c4 --no-online --no-batched --no-secondary --no-formats > gpurun_out/r5/rehearse2b.json 2> gpurun_out/r5/rehearse2b.err
echo "rc $?"; python3 -c "
import json; c=json.loads(open('gpurun_out/r5/rehearse2b.json').read().strip().splitlines()[-1]); print('sharded_c4:', c.get('sharded_c4'))"
