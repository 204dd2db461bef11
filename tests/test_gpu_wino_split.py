"""GPU parity of the split-operand Winograd kernel (csrc/conv3x3_wino_split.hip; st_set_conv_algo(ctx, 2)): the transform-domain
products of F(2x2,3x3) as six bf16 x bf16 partial products of three-way split fp32 operands on the bf16 matrix cores, fp32 accumulate.
fp32 results: the bars are the ones the IEEE-fp32 kernels are held to (tests/test_gpu_winograd.py: 1e-5 forward, 3e-5 data gradient
against the oracle; tests/test_gpu_fullsize.py at size) and, stated here because the arithmetic allows it, a tighter one: 2e-6 / 6e-6.
Reference: pycaffe Convolution forward / backward behind worker.py:84-86 and :100-106."""

import numpy as np
import pytest

import oracle
import style_transfer2_amd as st2
from helpers import rel_l2

pytestmark = pytest.mark.gpu
F32 = np.float32


def split_model(params, topo, **kw):
    m = st2.HipModel(params, topology=topo, **kw)
    m.engine.set_conv_algo(2)
    return m


@pytest.mark.parametrize('cin,cout,h,w', [
    (64, 64, 64, 96), (64, 128, 17, 32), (128, 128, 40, 72), (128, 256, 12, 12), (256, 256, 7, 44), (256, 512, 8, 8), (512, 512, 6, 4),
    (64, 64, 19, 40), (16, 64, 9, 36), (64, 192, 33, 100), (128, 64, 5, 64), (512, 512, 16, 16), (64, 64, 1, 4), (192, 128, 3, 8)])
def test_split_conv_forward_and_dgrad(cin, cout, h, w):
    """conv1_2 (cin -> cout) runs the split kernel forward (cin % 16 == 0, cout % 64 == 0) and, when cin % 64 == 0, backward
    (the (512, 512) cases at 16 x 16 and 6 x 4 also split K over workgroups).  Against the oracle, against the IEEE-fp32 Winograd
    kernel, and not bit-identical to it (the kernel really ran)."""
    topo = (('conv', 'conv1_1', 3, cin), ('conv', 'conv1_2', cin, cout))
    params = oracle.he_init_weights(topo, seed=cin + cout, bias_std=0.2)
    cpu = oracle.NetOracle(topo, params)
    gpu = split_model(params, topo)
    wino = st2.HipModel(params, topology=topo)
    rng = np.random.RandomState(h * w)
    x = (rng.randn(1, 3, h, w) * 40).astype(F32)
    fc = cpu.forward(x, ['conv1_2'])['conv1_2']
    fg = gpu.forward(x, ['conv1_1', 'conv1_2'])
    fw = wino.forward(x, ['conv1_2'])['conv1_2']
    assert rel_l2(fg['conv1_2'], fc) <= 2e-6, rel_l2(fg['conv1_2'], fc)
    assert rel_l2(fg['conv1_2'], fw) <= 2e-6
    assert not np.array_equal(fg['conv1_2'], fw)        # another arithmetic: the split kernel really ran
    d = rng.randn(*fc.shape).astype(F32)
    cpu.adopt_forward_state(fg)                         # same ReLU masks: compare the backward arithmetic only
    gc = cpu.backward({'conv1_2': d})
    gg = gpu.backward({'conv1_2': d})
    assert rel_l2(gg, gc) <= 6e-6, rel_l2(gg, gc)


@pytest.mark.parametrize('cin,cout,h,w', [(72, 200, 5, 64), (64, 128, 17, 33), (8, 96, 4, 32)])
def test_shapes_the_split_kernel_cannot_take_fall_back_to_the_fp32_winograd_kernel(cin, cout, h, w):
    """Channel counts / widths outside the kernel's requirements: algorithm 2 launches what algorithm 1 does, bit for bit."""
    topo = (('conv', 'conv1_1', 3, cin), ('conv', 'conv1_2', cin, cout))
    params = oracle.he_init_weights(topo, seed=cin + cout, bias_std=0.2)
    x = (np.random.RandomState(h * w).randn(1, 3, h, w) * 40).astype(F32)
    a, b = split_model(params, topo), st2.HipModel(params, topology=topo)
    fa, fb = a.forward(x, ['conv1_2'])['conv1_2'], b.forward(x, ['conv1_2'])['conv1_2']
    assert np.array_equal(fa, fb)
    d = np.random.RandomState(1).randn(*fa.shape).astype(F32)
    assert np.array_equal(a.backward({'conv1_2': d}), b.backward({'conv1_2': d}))


def test_split_chain_with_masks_injections_and_pools():
    """The data-gradient epilogue (ReLU mask from the blob below, injected diffs), the fused pools with their arg-max maps, the
    pool backward through the map (the split kernel has no unpooling input transform: maxpool_bwd_amap_k runs), split and
    IEEE-fp32 launches mixed in one chain (conv2_2's 192 outputs are no multiple of 64: forward on the fp32 kernel, backward split)."""
    topo = (('conv', 'conv1_1', 3, 64), ('conv', 'conv1_2', 64, 128), ('pool', 'pool1'),
            ('conv', 'conv2_1', 128, 128), ('conv', 'conv2_2', 128, 192), ('pool', 'pool2'), ('conv', 'conv3_1', 192, 64))
    params = oracle.he_init_weights(topo, seed=4, bias_std=0.2)
    cpu = oracle.NetOracle(topo, params)
    gpu = split_model(params, topo)
    rng = np.random.RandomState(3)
    for h, w in ((24, 40), (17, 72), (8, 8), (64, 96), (10, 52), (7, 4)):
        x = (rng.randn(1, 3, h, w) * 40).astype(F32)
        fc, fg = cpu.forward(x), gpu.forward(x)
        for name in fc:
            assert rel_l2(fg[name], fc[name]) <= 4e-6, (name, h, w, rel_l2(fg[name], fc[name]))
        cpu.adopt_forward_state(fg)
        for names in (['conv3_1'], ['conv3_1', 'pool2', 'conv2_2', 'conv2_1', 'pool1', 'conv1_2', 'conv1_1', 'data'], ['conv2_1'], ['conv1_2'], ['pool2']):
            diffs = {n: rng.randn(*fc[n].shape).astype(F32) for n in names}
            err = rel_l2(gpu.backward(diffs), cpu.backward(diffs))
            assert err <= 1e-5, (names, h, w, err)


@pytest.mark.parametrize('h,w', [(64, 96), (66, 100), (70, 256), (8, 32)])
def test_split_fused_pool_and_its_map_are_the_classic_pool_bit_for_bit(h, w, monkeypatch):
    """The fused 2x2 max-pool of the split kernel's epilogue and its one-byte arg-max map (conv1_2, conv2_2 of the VGG19 head): the
    pooled blobs equal the pool of the stored conv blob exactly, and the image gradient through the map equals the gradient through
    the classic pool backward (ST2_POOL_AMAP=0) bit for bit."""
    from oracle.caffe_net import maxpool_forward
    topo = oracle.VGG19_TOPOLOGY[:7]                    # conv1_1 conv1_2 pool1 conv2_1 conv2_2 pool2 conv3_1
    params = oracle.he_init_weights(topo, seed=3, bias_std=0.3)
    rng = np.random.RandomState(h + w)
    x = (rng.randn(1, 3, h, w) * 40).astype(F32)
    out = {}
    for amap in ('1', '0'):
        monkeypatch.setenv('ST2_POOL_AMAP', amap)
        gpu = split_model(params, topo)
        f = gpu.forward(x, ['conv1_2', 'pool1', 'conv2_1', 'conv2_2', 'pool2', 'conv3_1'])
        r2 = np.random.RandomState(7)
        diffs = {n: r2.randn(*f[n].shape).astype(F32) for n in ('conv3_1', 'conv2_1')}
        out[amap] = (f, gpu.backward(diffs), gpu.backward({'pool2': r2.randn(*f['pool2'].shape).astype(F32)}))
    f = out['1'][0]
    assert np.array_equal(f['pool1'][0], maxpool_forward(f['conv1_2'][0])[0])
    assert np.array_equal(f['pool2'][0], maxpool_forward(f['conv2_2'][0])[0])
    for n in f:
        assert np.array_equal(f[n], out['0'][0][n]), n
    assert np.array_equal(out['1'][1], out['0'][1]) and np.array_equal(out['1'][2], out['0'][2])
    assert float(np.abs(out['1'][1]).max()) > 0


@pytest.mark.parametrize('optimizer', ['adam', 'lbfgs'])
def test_split_lean_iterations_skip_dead_blobs_and_change_nothing(optimizer, monkeypatch):
    """Inside an iteration the full-resolution blob of a pooled, un-weighted layer is not written (the NOOUT build of the split kernel:
    conv1_2, conv2_2); ST2_LEAN32=0 writes everything: iterates and traces equal bit for bit.  256 x 512: no launch splits K."""
    topo = oracle.VGG19_TOPOLOGY[:10]                   # ... conv3_1 .. conv3_4
    params = oracle.he_init_weights(topo, seed=3, bias_std=0.2)
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (256, 512, 3)).astype(np.uint8), rs(2).randint(0, 256, (40, 36, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (256, 512, 3)).astype(np.uint8))
    weights = {'content': {'conv3_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1}, 'deepdream': {}}
    tv = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
    runs = {}
    for lean in ('1', '0'):
        monkeypatch.setenv('ST2_LEAN32', lean)
        dev = st2.StyleTransfer(split_model(params, topo))
        dev.set_input(init); dev.set_content(content); dev.set_style(style); dev.reset()
        dev.set_weights(weights, tv)
        dev.optimizer_cls = st2.AdamOptimizer if optimizer == 'adam' else st2.LBFGSOptimizer
        dev.set_step_size(10 if optimizer == 'adam' else 1)
        dev.reset()
        assert dev.start()
        out = [dev.step() for _ in range(3)]
        runs[lean] = [(np.asarray(i).copy(), dict(t)) for i, t in out]
        if lean == '1':
            with pytest.raises(st2.StError):
                dev.engine.get_blob('conv1_2')                                  # pooled, un-weighted: not written inside the step
        else:
            assert dev.engine.get_blob('conv1_2') is not None
    for (ia, ta), (ib, tb) in zip(runs['1'], runs['0']):
        assert np.array_equal(ia, ib)
        for k in ta:
            if k != 'time':
                assert ta[k] == tb[k] or (np.isnan(ta[k]) and np.isnan(tb[k])), k


def test_split_vgg19_objective_and_adam_steps_against_the_oracle():
    """worker.py:231-310 with every eligible conv on the split kernel: VGG19 to conv5_1 at 96 x 128, the headline's losses: first
    evaluation (loss, gradient, trace) and three Adam iterations against the CPU oracle at the bars of the fp32 engine
    (tests/test_gpu_parity.py: gradient rel-L2 1e-4, loss rtol 1e-4)."""
    from helpers import check_trace
    weights = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1}, 'deepdream': {}}
    tv = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
    topo = oracle.VGG19_TOPOLOGY
    params = oracle.he_init_weights(topo, seed=0)
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (96, 128, 3)).astype(np.uint8), rs(2).randint(0, 256, (80, 112, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (96, 128, 3)).astype(np.uint8))
    cpu = oracle.TransferOracle(oracle.NetOracle(topo, params, full_forward=False))
    model = st2.HipModel(params)
    model.engine.set_conv_algo(2)
    dev = st2.StyleTransfer(model)
    for st in (cpu, dev):
        st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
        st.set_weights(weights, tv)
    lo, go = cpu.opfunc(cpu.input)
    ld, gd = dev.opfunc()
    assert np.isclose(ld, lo, rtol=1e-4), (ld, lo)
    assert rel_l2(gd, go) <= 1e-4, rel_l2(gd, go)
    check_trace(list(cpu.traces[-1].data), list(cpu.traces[-1].data.values()), dev.traces[-1].data, rtol=1e-3)
    cpu.set_optimizer('adam', 10)
    dev.optimizer_cls = st2.AdamOptimizer; dev.set_step_size(10); dev.reset(); cpu.reset()
    assert cpu.start() and dev.start()
    for i in range(3):
        ic, tc = cpu.step()
        idv, td = dev.step()
        assert np.isclose(td['loss'], tc['loss'], rtol=2e-4), (i, td['loss'], tc['loss'])
    assert np.mean((np.asarray(idv) - ic) ** 2) <= 0.5
