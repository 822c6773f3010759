# the sharded block as a child job: world 1 through torchrun (must give a record), world 2 on a one-GPU box (must come back with an error, not hang)
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --sharded-child --gpus 1 --steps 5 2>/dev/null | tail -1 | cut -c1-400
python3 - <<'PY'
import sys, time, types
sys.path.insert(0, '.')
import bench
a = types.SimpleNamespace(steps=5, lm_iters=20)
t0 = time.time(); r = bench.run_sharded_child(a, 1, timeout_s=150); print('world 1 child: %.1f s' % (time.time() - t0), {k: r[k] for k in list(r)[:6]})
t0 = time.time(); r = bench.run_sharded_child(a, 2, timeout_s=90); print('world 2 child on one GPU: %.1f s' % (time.time() - t0), str(r)[:300])
PY
