"""What needs MORE THAN ONE device (the reference's unit of scale is one worker per `gpu` key, config.ini:10 +
router.py:67-84): skipped on a one-GPU box, run by the standard `pytest -m gpu` command the day two devices are visible.
Until then the grouped ncclSend / ncclRecv between two devices (engine_comm.cpp), ncclCommInitRank with world > 1 and the id rendezvous
across processes with distinct devices have never executed anywhere -- these tests are the first thing that does."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import style_transfer2_amd as st2

pytestmark = pytest.mark.gpu
F32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def device_count():
    """hipGetDeviceCount without initialising anything in THIS process that a child could inherit (torch.cuda.device_count() does not
    initialise the GPU on this image; a failure to import torch counts as one device)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:           # noqa: BLE001
        return 1


needs_two = pytest.mark.skipif(device_count() < 2, reason='needs >= 2 visible devices (this pool\'s boxes have one)')


def _reference(h, w, optimizer, steps):
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (h, w, 3)).astype(np.uint8), rs(2).randint(0, 256, (96, 80, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (h, w, 3)).astype(np.uint8))
    weights = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1}, 'deepdream': {}}
    ref = st2.StyleTransfer(st2.HipModel(oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0)))
    ref.set_input(init); ref.set_content(content); ref.set_style(style); ref.reset()
    ref.set_weights(weights, {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2})
    ref.optimizer_cls = st2.AdamOptimizer if optimizer == 'adam' else st2.LBFGSOptimizer
    ref.set_step_size(10 if optimizer == 'adam' else 1); ref.reset()
    assert ref.start()
    return [(np.asarray(i, F32).copy(), dict(t)) for i, t in (ref.step() for _ in range(steps))]


def _run_ranks(world, optimizer, steps, h, w, tmp_path):
    port = 29800 + (os.getpid() + (7 if optimizer == 'lbfgs' else 0) + 31 * world) % 150
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=REPO + os.pathsep + HERE + os.pathsep + os.environ.get('PYTHONPATH', ''))
    outs = [str(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'multi_device_child.py'), str(r), str(world), str(port), optimizer, str(steps),
                               str(h), str(w), outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0].decode(errors='replace'))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d exited with %s:\n%s' % (r, p.returncode, logs[r][-3000:])
    return [np.load(o) for o in outs]


def _compare(got, want, optimizer, steps):
    for step in range(steps):
        full = np.zeros_like(want[step][0])
        for g in got:
            y0, x0, y1, x1 = [int(v) for v in g['tile']]
            full[y0:y1, x0:x1] = g['images'][step]
            assert np.isclose(g['losses'][step], want[step][1]['loss'], rtol=1e-4 if step == 0 or optimizer == 'adam' else 2e-3), (step, g['losses'][step], want[step][1]['loss'])
            assert np.isclose(g['grads'][step], want[step][1]['grad'], rtol=1e-3 if step == 0 or optimizer == 'adam' else 1e-2), step
        mse = float(np.mean((full.astype(np.float64) - want[step][0]) ** 2))
        assert mse <= 1.0, (step, mse)


def test_the_rank_program_of_the_two_device_tests_runs_with_one_rank(tmp_path):
    """The child program of the tests below (tests/multi_device_child.py: fresh process, no torch, id rendezvous, st_comm_init, fused steps
    over real RCCL) on a 1 x 1 grid with ONE rank: runs on every box, so that the day two devices are visible the program itself is
    known to work and only the two-device transport is new."""
    got = _run_ranks(1, 'adam', 2, 96, 128, tmp_path)
    _compare(got, _reference(96, 128, 'adam', 2), 'adam', 2)


@needs_two
@pytest.mark.parametrize('optimizer', ['adam', 'lbfgs'])
def test_two_ranks_on_two_devices_over_real_rccl_match_the_plain_engine(optimizer, tmp_path):
    """VGG19 to conv5_1 on a 1 x 2 grid, one FRESH process per device: st_comm_unique_id -> socket rendezvous -> st_comm_init(world = 2),
    three fused steps (Adam / L-BFGS in its Gram form) with every all-reduce and strip exchange inside the engine over real RCCL; losses
    and the reassembled iterate against the plain engine on the whole image, at the bars of
    tests/test_gpu_parity.py::test_tiled_vgg19_two_ranks_match_single_gpu_engine (loss rtol 1e-4, gradient rms 1e-3, image MSE <= 1)."""
    h, w, steps, world = 176, 416, 3, 2
    _compare(_run_ranks(world, optimizer, steps, h, w, tmp_path), _reference(h, w, optimizer, steps), optimizer, steps)


@needs_two
def test_bench_fans_two_jobs_out_over_two_devices():
    """`bench.py --gpus 2` without --rehearse-one-gpu: two independent configs[1] jobs, one per device, no data-path collective
    (configs[3] at N = 2): n_gpus == 2 and a whole-job rate within 10 % of twice the N = 1 rate of the same short run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    common = ['--steps', '10', '--warmup', '3', '--repeats', '3', '--no-cpu-baseline', '--no-worker-level', '--no-extra-configs']
    lines = {}
    for n in (1, 2):
        res = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n)] + common, env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=900)
        assert res.returncode == 0, res.stderr.decode(errors='replace')[-3000:]
        lines[n] = json.loads([ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert lines[1]['n_gpus'] == 1 and lines[2]['n_gpus'] == 2
    assert 'rehearsal' not in lines[2]['config']
    ratio = lines[2]['value'] / lines[1]['value']
    assert 1.8 <= ratio <= 2.2, (lines[1]['value'], lines[2]['value'])
