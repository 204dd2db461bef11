"""The N > 1 path of the benchmark harness on the CPU: two gloo ranks, independent jobs, no collective
on the data path; only barrier + max/sum of scalars (BASELINE config 4 = "replicas of a job")."""
import os
import sys
import time

import pytest
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_main(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from style_transfer2_amd import distributed as d
    g = d.Group(backend='gloo')
    counter = {'n': 0}

    def step():                      # an independent "job step": rank 1 is the slow one
        counter['n'] += 1
        time.sleep(0.002 * (1 + rank))

    elapsed = d.timed_region(g, step, steps=10, warmup=2, sync=lambda: None)
    jobs = d.shard_jobs(5, world, rank)
    total_jobs = g.sum_over_ranks(len(jobs))
    out.put((rank, elapsed, counter['n'], jobs, total_jobs))
    g.close()


def test_two_rank_gloo_timed_region_takes_max_and_shards_jobs():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, e0, n0, j0, t0), (r1, e1, n1, j1, t1) = res
    assert n0 == n1 == 12                       # W untimed + exactly K timed steps on every rank
    assert e0 == pytest.approx(e1)              # both report the MAX over ranks
    assert e0 >= 10 * 0.004 * 0.9               # ... which is the slow rank's time
    assert j0 == [0, 2, 4] and j1 == [1, 3] and t0 == t1 == 5.0


def test_single_process_group_is_a_noop():
    sys.path.insert(0, REPO)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        os.environ.pop(k, None)
    from style_transfer2_amd import distributed as d
    g = d.Group()
    assert g.world == 1 and g.max_over_ranks(1.5) == 1.5 and g.sum_over_ranks(2) == 2.0
    g.barrier()
    assert d.shard_jobs(3, 1, 0) == [0, 1, 2]
    g.close()
