"""`bench.py --gpus N` must really start N ranks (reference scaling unit: one worker per `gpu` key, config.ini:10,
worker.py:328, router.py:67-84).  CPU test: the stub engine (a 1-ms sleep per step, no GPU), gloo, world size 2."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra)
    return env


def _run(args, env):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=env, cwd=REPO)


def test_gpus_2_without_a_launcher_spawns_two_ranks_and_prints_one_line():
    out = _run(['--gpus', '2', '--steps', '20', '--warmup', '2', '--repeats', '3', '--engine', 'stub'], _env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['jobs'] == 2 and d['steps'] == 20 and d['scaling'] == 'weak'
    # aggregate rate of two independent 1-ms-per-step jobs: about 2000 it/s, certainly more than one job's 1000
    assert 1000 < d['value'] < 2100, d['value']
    assert d['timing']['blocks'] == 3 and len(d['timing']['block_ms']) == 3
    assert d['timing']['ms_per_step']['min'] <= d['timing']['ms_per_step']['median'] <= d['timing']['ms_per_step']['max']
    assert abs(d['value'] - 2 * 1e3 / d['ms_per_step']) <= 1e-6 * d['value']


def test_launcher_world_size_must_match_gpus():
    out = _run(['--gpus', '1', '--steps', '2', '--engine', 'stub'],
               _env(RANK='0', LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999'))
    assert out.returncode == 2 and 'ranks' in out.stderr


def test_under_a_launcher_every_rank_joins(tmp_path):
    """The driver's way: torch.distributed.run starts the ranks; bench.py must not spawn again."""
    port = 29600 + os.getpid() % 300
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), BENCH, '--gpus', '2', '--steps', '10',
                          '--warmup', '1', '--repeats', '2', '--engine', 'stub'],
                         capture_output=True, text=True, timeout=300, env=_env(), cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    assert json.loads(lines[0])['n_gpus'] == 2


def test_a_rank_that_dies_early_ends_the_run_at_once_with_its_code():
    """One rank fails before the rendezvous (here: a test hook makes rank 1 exit with code 7): the launcher must not leave
    rank 0 waiting in the barrier for the collective's timeout."""
    import time
    t0 = time.time()
    out = _run(['--gpus', '2', '--steps', '5', '--engine', 'stub'], _env(ST2_BENCH_TEST_FAIL_RANK='1'))
    assert out.returncode == 7, (out.returncode, out.stderr[-1000:])
    assert 'rank 1 exited with code 7' in out.stderr
    assert time.time() - t0 < 60
