"""Multi-GPU plumbing for the job fan-out (BASELINE config 4): one process per GPU, independent
jobs, no data-path collective -- exactly how the reference scales (one app/worker pair per GPU,
config.ini:3-4,10; router.py:67-84).  torch.distributed is used only to line the ranks up around the
timed region and to take the max of their clocks: two scalars per block of steps.  That control plane
runs over gloo (host tensors) by default -- nothing on the data path crosses ranks, the process then
holds ONE HIP runtime (the engine's; torch.cuda is never initialised), and it is the path the tests
exercise; ST2_BENCH_BACKEND=nccl puts the same two collectives on RCCL.  The tile-sharded mode, whose
iteration does exchange data, owns its RCCL communicator inside the engine (csrc/engine_comm.cpp)."""

import os


def env_rank():
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def shard_jobs(n_jobs, world, rank):
    """Jobs (indices) owned by `rank`: round-robin, so every rank gets ceil or floor of n/world."""
    return list(range(rank, n_jobs, world))


class Group:
    """barrier() and max_over_ranks() for a weak-scaling timing harness; a no-op when world == 1."""

    def __init__(self, backend=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = None
        if self.world > 1:
            import datetime
            import torch
            import torch.distributed as dist
            # a rank that dies before the rendezvous must fail the others within minutes (bench.py's fan-out sets 120 s)
            timeout = datetime.timedelta(seconds=float(os.environ.get('ST2_RENDEZVOUS_TIMEOUT_S', '600')))
            backend = backend or os.environ.get('ST2_BENCH_BACKEND', 'gloo')
            if backend == 'nccl':
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device('cuda', self.local_rank)
                dist.init_process_group('nccl', device_id=self.device, timeout=timeout)
            else:
                self.device = torch.device('cpu')
                dist.init_process_group(backend, timeout=timeout)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def timed_region(group, step, steps, warmup, sync):
    """The benchmark contract: W untimed steps, then EXACTLY K steps bracketed by barrier + device sync
    on both sides; returns the max elapsed seconds over ranks."""
    import time
    for _ in range(warmup):
        step()
    sync()
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    group.barrier()
    return group.max_over_ranks(time.perf_counter() - t0)
