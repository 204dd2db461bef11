"""GPU parity of the Winograd F(2x2,3x3) conv3x3 kernel (csrc/conv3x3_winograd.hip) against the CPU oracle and
against the direct implicit-GEMM kernel.  Winograd computes the same fp32 products and sums in another
association (input / filter / output transforms), so the bar is a relative L2 error, stated per test:
1e-5 forward, 3e-5 data gradient -- the same bars the direct kernel is held to (test_gpu_parity.py)."""

import os

import numpy as np
import pytest

import oracle
import style_transfer2_amd as st2
from helpers import rel_l2

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(params=['auto', '0', '1', '3', '6', '8'])
def wino_cfg(request):
    """ST2_WINO_CFG: 0 = 128 channels x 4x32 pixels per workgroup, 1 = 64 channels x 8x32 pixels, 3 = 64 channels x 8x32
    pixels with the transform-domain positions split over the wave pairs (any-width shapes fall back to 1), 6 = 128 channels x
    4x32 pixels with eight waves, two per SIMD (any-width shapes fall back to 0), unset = chosen by
    shape (which may also split K over 2-4 workgroups when the launch would leave most CUs idle)"""
    old = os.environ.get('ST2_WINO_CFG')
    if request.param == 'auto':
        os.environ.pop('ST2_WINO_CFG', None)
    else:
        os.environ['ST2_WINO_CFG'] = request.param
    yield request.param
    if old is None:
        os.environ.pop('ST2_WINO_CFG', None)
    else:
        os.environ['ST2_WINO_CFG'] = old


@pytest.mark.parametrize('cin,cout,h,w', [
    (8, 96, 4, 32), (8, 128, 6, 8), (16, 128, 9, 36), (64, 128, 17, 32), (72, 200, 5, 64), (128, 128, 40, 72),
    (128, 256, 12, 12), (256, 256, 7, 44), (256, 512, 8, 8), (512, 512, 6, 4), (64, 160, 33, 100), (128, 128, 64, 96),
    (64, 64, 19, 40), (8, 48, 3, 4), (64, 64, 64, 96),
    # widths that are not multiples of 4: the any-width kernels (dword staging; 8-byte or, for odd W, 4-byte epilogue accesses)
    (64, 128, 17, 33), (128, 128, 40, 70), (512, 512, 9, 6), (16, 96, 5, 7), (64, 64, 12, 30), (72, 200, 6, 1), (128, 64, 9, 35)])
def test_winograd_conv_forward_and_dgrad(cin, cout, h, w, wino_cfg):
    """conv1_2 (cin -> cout) runs the Winograd kernel forward (M = cout >= 48) and, when cin >= 48, backward."""
    topo = (('conv', 'conv1_1', 3, cin), ('conv', 'conv1_2', cin, cout))
    params = oracle.he_init_weights(topo, seed=cin + cout, bias_std=0.2)
    cpu = oracle.NetOracle(topo, params)
    gpu = st2.HipModel(params, topology=topo)
    direct = st2.HipModel(params, topology=topo)
    direct.engine.set_conv_algo(False)
    rng = np.random.RandomState(h * w)
    x = (rng.randn(1, 3, h, w) * 40).astype(F32)
    fc = cpu.forward(x, ['conv1_2'])['conv1_2']
    fg = gpu.forward(x, ['conv1_1', 'conv1_2'])
    fd = direct.forward(x, ['conv1_2'])['conv1_2']
    assert rel_l2(fg['conv1_2'], fc) <= 1e-5, rel_l2(fg['conv1_2'], fc)
    assert rel_l2(fg['conv1_2'], fd) <= 1e-5
    assert not np.array_equal(fg['conv1_2'], fd)        # the Winograd kernel really ran (different association)
    d = rng.randn(*fc.shape).astype(F32)
    cpu.adopt_forward_state(fg)                         # same ReLU masks: compare the backward arithmetic only
    gc = cpu.backward({'conv1_2': d})
    gg = gpu.backward({'conv1_2': d})
    assert rel_l2(gg, gc) <= 3e-5, rel_l2(gg, gc)


def test_winograd_chain_with_masks_and_injections(wino_cfg):
    """Winograd dgrad epilogue: ReLU mask from the blob below, injected diffs, pools in between."""
    topo = (('conv', 'conv1_1', 3, 96), ('conv', 'conv1_2', 96, 128), ('pool', 'pool1'),
            ('conv', 'conv2_1', 128, 128), ('conv', 'conv2_2', 128, 192))
    params = oracle.he_init_weights(topo, seed=4, bias_std=0.2)
    cpu = oracle.NetOracle(topo, params)
    gpu = st2.HipModel(params, topology=topo)
    rng = np.random.RandomState(3)
    for h, w in ((24, 40), (17, 72), (8, 8), (19, 35), (10, 50), (7, 3)):
        x = (rng.randn(1, 3, h, w) * 40).astype(F32)
        fc, fg = cpu.forward(x), gpu.forward(x)
        for name in fc:
            assert rel_l2(fg[name], fc[name]) <= 1e-5, (name, h, w)
        cpu.adopt_forward_state(fg)
        for names in (['conv2_2'], ['conv2_2', 'conv2_1', 'pool1', 'conv1_2', 'conv1_1', 'data'], ['conv2_1'], ['conv1_2']):
            diffs = {n: rng.randn(*fc[n].shape).astype(F32) for n in names}
            err = rel_l2(gpu.backward(diffs), cpu.backward(diffs))
            assert err <= 3e-5, (names, h, w, err)


@pytest.mark.parametrize('h,w', [(64, 96), (32, 40), (66, 100), (48, 64), (64, 128), (70, 256), (8, 32)])
def test_pool_backward_through_the_arg_max_map_is_the_classic_one_bit_for_bit(h, w, monkeypatch):
    """A pool fused into its producing Winograd conv (aligned widths, one-tile-group builds) also leaves a one-byte arg-max map
    (first maximum of the stored blob, positive-after-bias flag), and the pool's backward routes the diff through it instead of
    re-reading the conv blob (maxpool_bwd_amap_k; ST2_POOL_AMAP=0 keeps maxpool_bwd_v4_k).  Where the width allows (W % 32 == 0) the
    data-gradient conv below the pool goes one further: it stages the POOLED diff and the map and unpools in its input transform
    (UNPOOL builds of the 128-channel and the half-tile kernel; ST2_WINO_UNPOOL=0 keeps the separate kernel).  Same routing rule and
    the same values into the same arithmetic, so the image gradient is identical bit for bit in all three forms -- through conv1_2
    (half-tile build) and conv2_2 (128-channel build), with diffs injected above and between."""
    topo = oracle.VGG19_TOPOLOGY[:7]                    # conv1_1 conv1_2 pool1 conv2_1 conv2_2 pool2 conv3_1
    params = oracle.he_init_weights(topo, seed=3, bias_std=0.3)
    rng = np.random.RandomState(h + w)
    x = (rng.randn(1, 3, h, w) * 40).astype(F32)
    out = {}
    for amap, unpool in (('1', '1'), ('1', '0'), ('0', '1')):
        monkeypatch.setenv('ST2_POOL_AMAP', amap)
        monkeypatch.setenv('ST2_WINO_UNPOOL', unpool)
        gpu = st2.HipModel(params, topology=topo)
        f = gpu.forward(x, ['pool1', 'conv2_1', 'pool2', 'conv3_1'])
        r2 = np.random.RandomState(7)
        diffs = {n: r2.randn(*f[n].shape).astype(F32) for n in ('conv3_1', 'conv2_1')}
        out[amap + unpool] = (f, gpu.backward(diffs), gpu.backward({'pool2': r2.randn(*f['pool2'].shape).astype(F32)}))
    for key in ('10', '01'):
        for n in out['11'][0]:
            assert np.array_equal(out['11'][0][n], out[key][0][n]), (key, n)
        assert np.array_equal(out['11'][1], out[key][1]), key
        assert np.array_equal(out['11'][2], out[key][2]), key
    assert float(np.abs(out['11'][1]).max()) > 0
    # ... and it is the oracle's gradient
    cpu = oracle.NetOracle(topo, params)
    cpu.forward(x, ['conv3_1'])
    cpu.adopt_forward_state(out['11'][0])
    cpu.adopt_forward_state(st2.HipModel(params, topology=topo).forward(x, ['conv1_1', 'conv1_2', 'conv2_2']))
    r2 = np.random.RandomState(7)
    diffs = {n: r2.randn(*out['11'][0][n].shape).astype(F32) for n in ('conv3_1', 'conv2_1')}
    assert rel_l2(out['11'][1], cpu.backward(diffs)) <= 3e-5


@pytest.mark.parametrize('h,w', [(64, 96), (66, 100), (64, 128), (70, 256), (8, 32)])
def test_big_tensor_builds_equal_the_plain_builds_bit_for_bit(h, w, monkeypatch):
    """Tensors of 4 GiB and more (one engine on an 8192 x 8192 image) take the BIG builds of the 128-channel and the half-tile kernel,
    plain and unpooling: the activation buffer resource is rebuilt per chunk at the chunk's own base (32-bit byte offsets stay below
    4 GiB), everything else is the same code.  ST2_WINO_FORCE_BIG=1 runs them at test size: forward blobs (with the fused pools and
    their arg-max maps) and the image gradient through conv1_2 (half tile), conv2_2 (128 channels) and the unpooling data gradients
    must equal the plain builds bit for bit.  The direct kernel's output check (32-bit ELEMENT offsets) is exercised by conv1_1; the
    64-bit-addressed style gradient (ST2_STYLE_FORCE_BIG=1) by test_big_style_gradient_kernel_matches_the_lds_dma_kernels."""
    topo = oracle.VGG19_TOPOLOGY[:7]                    # conv1_1 conv1_2 pool1 conv2_1 conv2_2 pool2 conv3_1
    params = oracle.he_init_weights(topo, seed=3, bias_std=0.3)
    rng = np.random.RandomState(h + w)
    x = (rng.randn(1, 3, h, w) * 40).astype(F32)
    out = {}
    for big in ('0', '1'):
        monkeypatch.setenv('ST2_WINO_FORCE_BIG', big)
        gpu = st2.HipModel(params, topology=topo)
        f = gpu.forward(x, ['conv1_2', 'pool1', 'conv2_1', 'conv2_2', 'pool2', 'conv3_1'])
        r2 = np.random.RandomState(7)
        diffs = {n: r2.randn(*f[n].shape).astype(F32) for n in ('conv3_1', 'conv2_1', 'conv1_2')}
        out[big] = (f, gpu.backward(diffs))
    for n in out['0'][0]:
        assert np.array_equal(out['0'][0][n], out['1'][0][n]), n
    assert np.array_equal(out['0'][1], out['1'][1])
    assert float(np.abs(out['0'][1]).max()) > 0


@pytest.mark.parametrize('optimizer,weighted_pooled', [('adam', False), ('lbfgs', False), ('adam', True)])
def test_lean_fp32_iterations_skip_dead_blobs_and_change_nothing(optimizer, weighted_pooled, monkeypatch):
    """Inside an iteration (st_step) the full-resolution blob of a pooled layer that carries no weight is dead: the next conv reads the
    pooled blob, the pool's backward the arg-max map (ReLU sign included).  The NOOUT builds of the half-tile (conv1_2) and the
    128-channel kernel (conv2_2) do not write it (conv1_2 at 1024^2: 268 MB per step and the row exchange / bias / ReLU / store of
    every accumulator row).  ST2_LEAN32=0 writes everything: the iterates and traces must be equal bit for bit; st_opfunc (the test
    hook) always writes every blob; a weighted pooled layer keeps its blob.  256 x 512: large enough that neither launch splits K
    (a split-K launch has no fused pool)."""
    topo = oracle.VGG19_TOPOLOGY[:10]                   # ... conv3_1 .. conv3_4
    params = oracle.he_init_weights(topo, seed=3, bias_std=0.2)
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (256, 512, 3)).astype(np.uint8), rs(2).randint(0, 256, (40, 36, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (256, 512, 3)).astype(np.uint8))
    weights = {'content': {'conv3_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1}, 'deepdream': {}}
    if weighted_pooled:
        weights['style']['conv2_2'] = 0.5
    tv = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
    runs = {}
    for lean in ('1', '0'):
        monkeypatch.setenv('ST2_LEAN32', lean)
        dev = st2.StyleTransfer(st2.HipModel(params, topology=topo))
        dev.set_input(init); dev.set_content(content); dev.set_style(style); dev.reset()
        dev.set_weights(weights, tv)
        dev.optimizer_cls = st2.AdamOptimizer if optimizer == 'adam' else st2.LBFGSOptimizer
        dev.set_step_size(10 if optimizer == 'adam' else 1)
        dev.reset()
        assert dev.start()
        out = [dev.step() for _ in range(4)]
        runs[lean] = [(np.asarray(i).copy(), dict(t)) for i, t in out]
        if lean == '1':
            with pytest.raises(st2.StError):
                dev.engine.get_blob('conv1_2')                                  # pooled, un-weighted: not written inside the step
            if weighted_pooled:
                assert dev.engine.get_blob('conv2_2') is not None               # pooled but weighted: kept
            else:
                with pytest.raises(st2.StError):
                    dev.engine.get_blob('conv2_2')
        else:
            assert dev.engine.get_blob('conv1_2') is not None
        dev.opfunc()
        assert dev.engine.get_blob('conv1_2') is not None                       # the test hook writes every blob
    for (ia, ta), (ib, tb) in zip(runs['1'], runs['0']):
        assert np.array_equal(ia, ib)
        for k in ta:
            if k != 'time':
                assert ta[k] == tb[k] or (np.isnan(ta[k]) and np.isnan(tb[k])), k


@pytest.mark.parametrize('h,w', [(64, 96), (66, 100), (70, 256), (8, 32), (17, 72)])
def test_specialised_epilogues_equal_the_generic_epilogue_bit_for_bit(h, w, monkeypatch):
    """Round 5: the 128-channel and the half-tile build have an epilogue per launch KIND (forward, forward + pool, pooled blob only,
    data gradient with / without the ReLU mask, unpooling data gradient) -- buffer accesses with zero-size resources and out-of-range
    offsets instead of run-time flags, branches and 64-bit addresses; ST2_WINO_EPI=0 keeps the generic epilogue.  The arithmetic is
    the same, operation for operation: forward blobs (fused pools and their arg-max maps included: the pooled blobs and the gradient
    through the maps), the image gradient with diffs injected at conv blobs with and without a mask below them, and lean iterations
    (pooled-blob-only builds) must be the same bits."""
    topo = oracle.VGG19_TOPOLOGY[:7]                    # conv1_1 conv1_2 pool1 conv2_1 conv2_2 pool2 conv3_1
    params = oracle.he_init_weights(topo, seed=3, bias_std=0.3)
    rng = np.random.RandomState(h + w)
    x = (rng.randn(1, 3, h, w) * 40).astype(F32)
    out = {}
    for epi in ('1', '0'):
        monkeypatch.setenv('ST2_WINO_EPI', epi)
        gpu = st2.HipModel(params, topology=topo)
        f = gpu.forward(x, ['conv1_1', 'conv1_2', 'pool1', 'conv2_1', 'conv2_2', 'pool2', 'conv3_1'])
        r2 = np.random.RandomState(7)
        grads = [gpu.backward({n: r2.randn(*f[n].shape).astype(F32) for n in names})
                 for names in (('conv3_1', 'conv2_1'), ('pool2',), ('conv3_1', 'conv2_2', 'conv2_1', 'pool1', 'conv1_2', 'conv1_1'), ('conv2_2',))]
        out[epi] = (f, grads)
    for n in out['1'][0]:
        assert np.array_equal(out['1'][0][n], out['0'][0][n]), n
    for ga, gb in zip(out['1'][1], out['0'][1]):
        assert np.array_equal(ga, gb)
    assert float(np.abs(out['1'][1][0]).max()) > 0


def test_specialised_epilogues_in_lean_iterations_change_nothing(monkeypatch):
    """... and inside iterations (the pooled-blob-only builds, fused Adam): iterates and traces of three steps, bit for bit."""
    topo = oracle.VGG19_TOPOLOGY[:10]
    params = oracle.he_init_weights(topo, seed=3, bias_std=0.2)
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (256, 512, 3)).astype(np.uint8), rs(2).randint(0, 256, (40, 36, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (256, 512, 3)).astype(np.uint8))
    weights = {'content': {'conv3_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1}, 'deepdream': {}}
    runs = {}
    for epi in ('1', '0'):
        monkeypatch.setenv('ST2_WINO_EPI', epi)
        dev = st2.StyleTransfer(st2.HipModel(params, topology=topo))
        dev.set_input(init); dev.set_content(content); dev.set_style(style); dev.reset()
        dev.set_weights(weights, {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2})
        dev.optimizer_cls = st2.AdamOptimizer; dev.set_step_size(10); dev.reset()
        assert dev.start()
        runs[epi] = [(np.asarray(i).copy(), dict(t)) for i, t in (dev.step() for _ in range(3))]
    for (ia, ta), (ib, tb) in zip(runs['1'], runs['0']):
        assert np.array_equal(ia, ib)
        for k in ta:
            if k != 'time':
                assert ta[k] == tb[k] or (np.isnan(ta[k]) and np.isnan(tb[k])), k
