"""SURVEY section 8(f) rows 1 and 4 on the real engine (``pytest -m gpu``): a worker configured like the reference's
(config.ini:28-29 ``prototxt`` / ``caffemodel``; worker.py:58-61 ``caffe.Net(prototxt, 1, weights=...)``) builds its model
from a network definition + a protobuf weight file and iterates; headless jobs prepare their images the way the web app
does (utils.py:210-229, app.py:82,244-262)."""
import configparser
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import style_transfer2_amd as st2
from style_transfer2_amd import caffemodel, jobs, prototxt, weights as st2_weights
import worker as worker_mod
from helpers import GOLDEN

pytestmark = pytest.mark.gpu
F32 = np.float32
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOPO = (('conv', 'conv1_1', 3, 8), ('conv', 'conv1_2', 8, 8), ('pool', 'pool1'), ('conv', 'conv2_1', 8, 16))
WEIGHTS = {'content': {'conv2_1': 0.08}, 'style': {'conv1_1': 1, 'conv1_2': 1, 'conv2_1': 1}, 'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}


def _config(tmp_path, **keys):
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = {k: str(v) for k, v in keys.items()}
    return cp['DEFAULT']


def _run3(tr):
    rs = np.random.RandomState
    tr.set_input(rs(3).randint(0, 256, (24, 32, 3)).astype(np.uint8))
    tr.set_content(rs(1).randint(0, 256, (24, 32, 3)).astype(np.uint8))
    tr.set_style(rs(2).randint(0, 256, (20, 20, 3)).astype(np.uint8))
    tr.reset()
    tr.set_weights(WEIGHTS, PARAMS)
    tr.optimizer_cls = st2.AdamOptimizer; tr.set_step_size(10); tr.reset()
    assert tr.start()
    return [tr.step() for _ in range(3)]


@pytest.mark.parametrize('fixture', ['caffemodel_v2_shape_packed.bin', 'caffemodel_v1_legacy_unpacked.bin'])
@pytest.mark.parametrize('is_bgr', [False, True])
def test_worker_builds_its_model_from_prototxt_and_caffemodel(tmp_path, fixture, is_bgr):
    """build_transfer on `prototxt = ...` + `caffemodel = ....caffemodel` (bytes serialized by google.protobuf, tests/golden) against
    the same weights handed over as .npz: identical iterates and traces, with and without the BGR -> RGB flip of the first layer."""
    params = st2_weights.he_normal(TOPO, seed=4, bias_std=0.3)              # what the golden fixtures hold
    (tmp_path / 'net.prototxt').write_text(prototxt.write(TOPO))
    (tmp_path / 'net.caffemodel').write_bytes(open(os.path.join(GOLDEN, fixture), 'rb').read())
    expect = {k: (w[:, ::-1].copy() if (is_bgr and k == 'conv1_1') else w, b) for k, (w, b) in params.items()}
    st2_weights.save_npz(str(tmp_path / 'net.npz'), expect)
    via_caffe = worker_mod.build_transfer(_config(tmp_path, gpu=0, prototxt=tmp_path / 'net.prototxt', caffemodel=tmp_path / 'net.caffemodel',
                                                  caffemodel_is_bgr='yes' if is_bgr else 'no'))
    via_npz = worker_mod.build_transfer(_config(tmp_path, gpu=0, prototxt=tmp_path / 'net.prototxt', caffemodel=tmp_path / 'net.npz'))
    assert via_caffe.model.layers() == ['data', 'conv1_1', 'conv1_2', 'pool1', 'conv2_1'] == via_npz.model.layers()
    for (ia, ta), (ib, tb) in zip(_run3(via_caffe), _run3(via_npz)):
        assert np.array_equal(ia, ib)
        assert {k: v for k, v in ta.items() if k != 'time'} == {k: v for k, v in tb.items() if k != 'time'}
    # ... and it is the network the oracle computes with those weights
    cpu = oracle.TransferOracle(oracle.NetOracle(TOPO, expect))
    rs = np.random.RandomState
    cpu.set_input(rs(3).randint(0, 256, (24, 32, 3)).astype(np.uint8)); cpu.set_content(rs(1).randint(0, 256, (24, 32, 3)).astype(np.uint8))
    cpu.set_style(rs(2).randint(0, 256, (20, 20, 3)).astype(np.uint8)); cpu.reset()
    cpu.set_weights(WEIGHTS, PARAMS); cpu.set_optimizer('adam', 10)
    assert cpu.start()
    fresh = worker_mod.build_transfer(_config(tmp_path, gpu=0, prototxt=tmp_path / 'net.prototxt', caffemodel=tmp_path / 'net.caffemodel',
                                              caffemodel_is_bgr='yes' if is_bgr else 'no'))
    for (img, tr), _ in zip(_run3(fresh), range(3)):
        ic, tc = cpu.step()
        assert np.isclose(tr['loss'], tc['loss'], rtol=1e-4) and np.mean((img - ic) ** 2) <= 1e-2


def test_worker_refuses_a_definition_it_cannot_run(tmp_path, capsys):
    (tmp_path / 'net.prototxt').write_text(prototxt.write(TOPO).replace('kernel_size: 3', 'kernel_size: 5', 1))
    with pytest.raises(SystemExit) as e:
        worker_mod.build_transfer(_config(tmp_path, gpu=0, prototxt=tmp_path / 'net.prototxt', weights='synthetic'))
    assert e.value.code == 2 and 'not a 3x3' in capsys.readouterr().err


def _example_pair():
    from PIL import Image
    src = np.load(os.path.join(GOLDEN, 'config1_sources.npz'))
    return Image.fromarray(src['golden_gate']), Image.fromarray(src['starry_night'])


def test_run_job_prepares_images_like_the_app_and_matches_the_oracle():
    """jobs.run_job on the reference's example pair: resize_to_fit(256) gives the reference's own arrays (config1_inputs.npz), the
    job then runs 10 Adam iterations on the device; the CPU oracle fed with the same arrays follows (reference app.py:244-262)."""
    content, style = _example_pair()
    ref = np.load(os.path.join(GOLDEN, 'config1_inputs.npz'))
    assert np.array_equal(np.uint8(jobs.resize_to_fit(content, 256)), ref['golden_gate'])
    assert np.array_equal(np.uint8(jobs.resize_to_fit(style, 256)), ref['starry_night'])
    params = oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0)
    seen = []
    dev = st2.StyleTransfer(st2.HipModel(params))
    image = jobs.run_job(dev, content, style, 10, size=256, optimizer='adam', seed=5, callback=lambda i, img, tr: seen.append((i, tr['loss'])))
    assert image.shape == (192, 256, 3) and image.dtype == F32 and [i for i, _ in seen] == list(range(1, 11))
    quiet = jobs.run_job(st2.StyleTransfer(st2.HipModel(params)), content, style, 10, size=256, optimizer='adam', seed=5)
    assert np.array_equal(quiet, image)                                     # the device-resident loop (no per-step read-back) is the same job
    cpu = oracle.TransferOracle(oracle.NetOracle(oracle.VGG19_TOPOLOGY, params, full_forward=False))
    cpu.set_input(jobs.noise_image((192, 256), 5)); cpu.set_content(ref['golden_gate']); cpu.set_style(ref['starry_night']); cpu.reset()
    cpu.set_weights(jobs.DEFAULT_WEIGHTS, jobs.DEFAULT_PARAMS); cpu.set_optimizer('adam', 10)
    assert cpu.start()
    for i in range(10):
        ic, tc = cpu.step()
        assert np.isclose(seen[i][1], tc['loss'], rtol=2e-4), (i, seen[i][1], tc['loss'])
    assert np.mean((image - ic) ** 2) <= 0.25                               # 0..255 units; Adam's first steps are sign-like (measured 0.055 after 20)


def test_stylize_cli_and_bench_examples_mode(tmp_path):
    """tools/stylize.py end to end on files, and `bench.py --examples` (BASELINE configs[0]): one JSON line whose inputs went through
    jobs.resize_to_fit, with the CPU oracle's 50 iterations timed beside the device's and the final images compared."""
    content, style = _example_pair()
    content.save(tmp_path / 'c.png'); style.save(tmp_path / 's.png')
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'stylize.py'), str(tmp_path / 'c.png'), str(tmp_path / 's.png'),
                          str(tmp_path / 'o.png'), '--size', '128', '--iters', '5'], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-1500:]
    from PIL import Image
    assert Image.open(tmp_path / 'o.png').size == (128, 96)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--examples', '--examples-iters', '8', '--steps', '10', '--warmup', '2', '--repeats', '2',
                          '--no-worker-level'], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-1500:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['config']['workload'].startswith('configs[0]') and d['config']['resize_to_fit_matches_reference_fixture'] is True
    assert d['cpu_baseline']['kind'] == 'port' and '8 adam' in d['cpu_baseline']['sample'] and d['cpu_baseline']['value'] > 0
    assert d['parity']['image_after'].startswith('8 adam') and d['parity']['image_mse'] <= 0.5 and d['parity']['step_loss_rel'] <= 1e-3
    assert d['value'] > 50 * d['cpu_baseline']['value']


def test_run_tiled_job_on_one_gpu_equals_run_job():
    """jobs.run_tiled_job: the example pair at 256 px cut 2 x 2 into tiles that all live on the one GPU (tiled.InProcessFabric,
    device-to-device exchanges) against jobs.run_job on the whole image: the same ten Adam iterations."""
    content, style = _example_pair()
    params = oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0)
    whole = jobs.run_job(st2.StyleTransfer(st2.HipModel(params)), content, style, 10, size=256, optimizer='adam', seed=5)
    seen = []
    tiles = jobs.run_tiled_job(params, content, style, 10, (2, 2), size=256, seed=5, callback=lambda i, vals: seen.append((i, vals[-2])))
    assert tiles.shape == whole.shape == (192, 256, 3) and tiles.dtype == F32 and [i for i, _ in seen] == list(range(1, 11))
    mse = float(np.mean((tiles.astype(np.float64) - whole) ** 2))
    print('[tiled job 2x2 at 192x256] image MSE %.3g against the whole-image job after 10 Adam iterations' % mse)
    assert mse <= 0.05 and np.isfinite(seen[-1][1])


def test_run_tiled_job_with_lbfgs_on_one_gpu_equals_run_job():
    """The L-BFGS variant over the sharded image (every utils.dot a per-rank partial + one all-reduce, tiled.TiledTransfer) with its
    ranks as threads of this process (tiled.LocalComm): four steps -- while the fixed-step iteration contracts -- against run_job."""
    content, style = _example_pair()
    params = oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0)
    whole = jobs.run_job(st2.StyleTransfer(st2.HipModel(params)), content, style, 4, size=256, optimizer='lbfgs', seed=5)
    tiles = jobs.run_tiled_job(params, content, style, 4, (2, 2), size=256, seed=5, optimizer='lbfgs')
    mse = float(np.mean((tiles.astype(np.float64) - whole) ** 2))
    print('[tiled L-BFGS job 2x2 at 192x256] image MSE %.3g against the whole-image job after 4 steps' % mse)
    assert tiles.shape == whole.shape and mse <= 0.05
