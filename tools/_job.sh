cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1H; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/r1H/build.log 2>&1
echo "== torchrun world=1"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-220
echo "== 2 ranks on one GPU, gloo, restarts, scale 0.2"; SKF_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.2 2>&1 | grep -E '^\{|Error|error' | cut -c1-300
echo "== 2 ranks on one GPU, gloo, relations, scale 0.2"; SKF_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.2 --mode relations 2>&1 | grep -E '^\{|Error|error' | cut -c1-300
echo "== 1 rank nccl init path (world 1 forced through dist)"; timeout 300 python - <<'PY'
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29514', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
t = torch.ones(1024, device='cuda'); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print('nccl world-1 all_reduce ok', float(t.sum()))
dist.destroy_process_group()
PY
