"""Rehearsal of bench.LineWatchdog in a process that holds live GPU state: a one-rank RCCL process group of torch.distributed, an
RCCL communicator of the library (skf_comm_create), device buffers -- fork, keep computing and reducing, disarm.  The forked
child must neither disturb the parent's GPU work nor print when the parent comes back:
    python tools/rehearse_watchdog.py          (on the GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                     # noqa: E402
import torch.distributed as dist                 # noqa: E402
import bench                                     # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29531')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.ones(1 << 20, device='cuda')
dist.all_reduce(x)
torch.cuda.synchronize()
w = bench.run_workload('c3', 'bf16', 3, 1, 0.1)             # a plan of the library before the fork
dog = bench.LineWatchdog(json.dumps({'value': 0, 'strong': {'error': 'rank 0 died inside the strong leg'}}))
w2 = bench.run_workload('c3', 'bf16', 3, 1, 0.1, 'uniform', 'owned', 0, 1, dist, 'nccl')   # one-rank RCCL communicator of the library
y = torch.arange(1 << 20, device='cuda', dtype=torch.float32)
dist.all_reduce(y)
torch.cuda.synchronize()
assert float(y[12345]) == 12345.0 and float(x[7]) == 1.0
dog.disarm()
print(json.dumps({'value': 1, 'before_fork_it_s': 3 / w['elapsed'], 'after_fork_owned_it_s': 3 / w2['elapsed'], 'comm': w2['comm']}), flush=True)
dist.destroy_process_group()
