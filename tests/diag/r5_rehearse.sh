cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --rehearse-gloo --batch-queue 0 > gpurun_out/r5/rehearse2.json 2> gpurun_out/r5/rehearse2.err
echo "rc $?"; tail -c 1500 gpurun_out/r5/rehearse2.json; echo; grep -v "^{" gpurun_out/r5/rehearse2.err | tail -5
