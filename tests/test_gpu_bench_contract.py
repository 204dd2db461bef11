"""The bench.py contract on the GPU box: one JSON line with the keys the driver and the judge read (a small
image and a handful of steps: this checks the plumbing, not the speed)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    env = dict(os.environ)
    env.pop('RANK', None); env.pop('WORLD_SIZE', None); env.pop('LOCAL_RANK', None)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2',
                          '--size', '256'] + list(extra), capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout          # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    d = run_bench('--cpu-size', '64')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2
    assert d['unit'] == 'it/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config']
    assert abs(d['value'] - 1e3 / d['ms_per_step']) <= 1e-6 * d['value']
    assert 'custom' in d['config']['workload']                  # 256 px is not a BASELINE.json config and says so
    assert d['timing']['blocks'] == 5 and len(d['timing']['block_ms']) == 5
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'algorithmic', 'algorithmic_over_peak'):
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == pytest.approx(157.3)
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'])
    assert 0 < r['frac'] < 1                                    # a fraction of the MFMA peak: executed flops
    assert 0 < r['achieved'] <= r['algorithmic']
    assert r['traffic'] is None and r['traffic_source']         # no PMC pass for this configuration: null, and says why
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'it/s' and c['value'] > 0 and c['cores'] >= 1 and '64x64' in c['sample']
    assert d['parity'] is None                                  # the CPU sample was taken at another size
    w = d['worker_level']
    assert w['sync_iterate_it_s'] > 0 and w['async_iterate_it_s'] > 0 and w['iterate_MB'] == pytest.approx(256 * 256 * 3 * 4 / 1e6, rel=0.05)


def test_bench_parity_leg_at_the_benched_size():
    d = run_bench('--size', '128', '--no-worker-level')
    p = d['parity']
    assert p['loss_rel'] <= 1e-4 and p['grad_rel_l2'] <= 5e-3 and p['grad_cosine'] >= 0.9999
    assert p['relu_sign_flips'] <= 50 and p['activations'] > 1e6


def test_bench_bf16_and_lbfgs_variants_run():
    d = run_bench('--precision', 'bf16', '--optimizer', 'lbfgs', '--no-cpu-baseline', '--repeats', '2')
    assert 'bf16' in d['dtype'] and 'cpu_baseline' not in d and d['timing']['blocks'] == 2
    assert d['roofline']['peak'] == pytest.approx(2516.6)


def test_bench_gpus_fan_out_rehearsed_on_one_gpu():
    """`--gpus 2` without a launcher starts two fresh rank processes before anything touches the GPU; rank 0 prints the one line
    with n_gpus = 2 and the aggregate rate.  Here both ranks share the box's only card (`--rehearse-one-gpu`: device 0, gloo
    rendezvous), so this checks the plumbing the driver's 2/4/8-GPU runs go through, not a rate."""
    d = run_bench('--gpus', '2', '--rehearse-one-gpu', '--size', '128', '--no-cpu-baseline', '--no-worker-level', '--repeats', '2')
    assert d['n_gpus'] == 2 and d['config']['jobs'] == 2 and d['scaling'] == 'weak'
    assert 'rehearsal' in d['config'] and d['value'] > 0
    assert d['value'] == pytest.approx(2 / (d['ms_per_step'] * 1e-3), rel=1e-6)      # whole-job aggregate: two ranks' steps per unit of time


def test_bench_tiled_grid_on_one_gpu_runs_every_rank_in_process():
    """`bench.py --tiled RxC` with one GPU: every rank of the grid is an engine context of the one process (tools/bench_tiled_one_gpu.py,
    tiled.InProcessFabric) -- how a single card runs an image no single engine holds; here a 2 x 2 grid over 512 x 512."""
    d = run_bench('--tiled', '2x2', '--size', '512')
    assert d['ranks'] == 4 and d['n_gpus'] == 1 and d['grid'] == '2x2' and d['value'] > 0
    assert d['messages_per_step'] > 0 and 'device-to-device' in d['config']['transport'] and d['loss'] > 0
