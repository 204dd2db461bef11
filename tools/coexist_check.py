#!/usr/bin/env python3
"""bench.py's multi-GPU order of events on ONE GPU: torch.cuda + an RCCL process group (world size 1) come up first,
then the engine (its own HIP runtime copy: /opt/rocm's, torch bundles another) runs steps, then a collective.
Run on the GPU box:  python tools/coexist_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ['RANK'] = '0'; os.environ['LOCAL_RANK'] = '0'; os.environ['WORLD_SIZE'] = '1'
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dist.barrier()
print('RCCL group up', flush=True)
import bench
from style_transfer2_amd import distributed as st2_dist
job = bench.make_job(bench.images(512) + (bench.WEIGHTS, bench.PARAMS), 'adam', 0)
t0 = time.perf_counter()
for _ in range(10):
    job.step_async()
job.engine.sync()
print('10 engine steps after torch/RCCL init: %.1f ms' % (1e3 * (time.perf_counter() - t0)), flush=True)
t = torch.tensor([1.5], dtype=torch.float64, device='cuda:0')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print('all_reduce ->', float(t.item()), flush=True)
image, trace = job.step()
print('iterate', image.shape, 'loss', trace['loss'], flush=True)
dist.barrier()
dist.destroy_process_group()
print('OK', flush=True)
