"""Parity at BASELINE.json's own sizes: configs[1] (1024 x 1024, fp32, Adam) and configs[2] (2048 x 2048, bf16 conv
operands, L-BFGS).  ``pytest -m gpu``; every comparison is HIP (through the C ABI) against the CPU oracle on the
bench's exact inputs (bench.images / he_normal seed 0 / bench.WEIGHTS / bench.PARAMS).

Three kinds of check, because ReLU and max-pool are discontinuous (oracle.NetOracle.adopt_forward_state explains):
  1. every conv shape the two configs launch, as a layer of its own at its production size (K, M, H, W identical to
     the launch inside the VGG19 step: same tile configuration, split-K factor, fused pool, 32-bit offsets), forward and
     data gradient on a shared forward state -- tight (fp32: summation order only; bf16: rounded-operand oracle fed
     with the GPU's own input blob);
  2. the whole VGG19 objective at full size: forward blobs of the six weighted layers, loss, trace, the ranged
     backward on the GPU's own forward state -- tight;
  3. the end-to-end gradient, with the number of ReLU-sign / pool-arg-max flips between the two forwards and the
     fraction of pixels they touch counted, reported (gpurun_out/parity_fullsize.json) and bounded.
The numbers measured on MI355X are recorded in DESIGN.md section 5.
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle.caffe_net import bf16_round, conv3x3_forward, maxpool_forward
import style_transfer2_amd as st2
from style_transfer2_amd import weights as st2_weights
from helpers import rel_l2, check_trace, trained_like_weights, receptive_geometry, paint_receptive_fields

pytestmark = pytest.mark.gpu
F32 = np.float32
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.environ.get('ST2_PARITY_REPORT') or os.path.join(REPO, 'gpurun_out', 'parity_fullsize.json')

WEIGHTS = {'content': {'conv4_2': 0.08},
           'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1},
           'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
WEIGHTED = ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv4_2', 'conv5_1']
TO_CONV5_1 = [l[1] for l in oracle.VGG19_TOPOLOGY[:17]]


def report(key, values):
    """Merge measured numbers into gpurun_out/parity_fullsize.json (read back by the builder, cited in DESIGN.md)."""
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        data = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        data[key] = values
        json.dump(data, open(REPORT, 'w'), indent=1, sort_keys=True)
    except OSError:
        pass
    print('[parity] %s: %s' % (key, json.dumps(values)))


def images(size):
    rs = np.random.RandomState
    shape = (size, size, 3)
    return (rs(1).randint(0, 256, shape).astype(np.uint8), rs(2).randint(0, 256, shape).astype(np.uint8),
            rs(3).randint(0, 256, shape).astype(np.uint8))


def test_the_bench_inputs_are_the_ones_checked_here():
    import bench
    assert bench.WEIGHTS == WEIGHTS and bench.PARAMS == PARAMS
    for a, b in zip(bench.images(64), images(64)):
        assert np.array_equal(a, b)
    pa, pb = st2_weights.he_normal(st2.VGG19_TOPOLOGY[:4], seed=0), oracle.he_init_weights(oracle.VGG19_TOPOLOGY[:4], seed=0)
    for name in pa:
        assert np.array_equal(pa[name][0], pb[name][0]) and np.array_equal(pa[name][1], pb[name][1])


# ------------------------------------------------------------------------------ 1. production layer shapes
# (K, M, edge at 1024^2, pooled afterwards?) of every conv the step launches at 1024^2; x2 edge for the 2048^2 config
VGG_SHAPES = [
    ('conv1_2', 64, 64, 1024, True), ('conv2_1', 64, 128, 512, False), ('conv2_2', 128, 128, 512, True),
    ('conv3_1', 128, 256, 256, False), ('conv3_2', 256, 256, 256, True), ('conv4_1', 256, 512, 128, False),
    ('conv4_2', 512, 512, 128, True), ('conv5_1', 512, 512, 64, False)]


def _layer_case(K, M, edge, pooled, precision, rows=None):
    """A three-layer network conv_a (3 -> K), conv_b (K -> M) [, pool] at rows x edge (rows = edge unless given): conv_b is the production shape."""
    rows = rows or edge
    topo = (('conv', 'conv_a', 3, K), ('conv', 'conv_b', K, M)) + ((('pool', 'pool_b'),) if pooled else ())
    params = oracle.he_init_weights(topo, seed=K + M, bias_std=0.2)
    gpu = st2.HipModel(params, topology=topo, precision=precision)
    cpu = oracle.NetOracle(topo, params, operands='bf16' if precision == 'bf16' else 'fp32')
    x = (np.random.RandomState(edge).randn(1, 3, rows, edge) * 40).astype(F32)
    return topo, params, gpu, cpu, x


@pytest.mark.parametrize('name,K,M,edge,pooled', VGG_SHAPES)
def test_production_conv_shapes_fp32_at_1024(name, K, M, edge, pooled):
    topo, params, gpu, cpu, x = _layer_case(K, M, edge, pooled, 'fp32')
    names = [l[1] for l in topo]
    fg = gpu.forward(x, names)
    fc = cpu.forward(x, names)
    tol = 2e-6 if K <= 256 else 4e-6            # fp32 k-chain of 9 K terms vs BLAS blocking: ~sqrt(K) eps (measured 3e-7 .. 9e-7)
    errs = {n: rel_l2(fg[n], fc[n]) for n in names}
    flips = int(np.sum((fg['conv_b'] > 0) != (fc['conv_b'] > 0)))
    for n in names:
        assert errs[n] <= tol, (n, errs)
    assert flips <= 1e-5 * fc['conv_b'].size + 5, flips
    # data gradient with injections at every blob, on the GPU's forward state (masks / arg-max shared)
    cpu.adopt_forward_state(fg)
    rng = np.random.RandomState(7)
    diffs = {n: rng.randn(*fg[n].shape).astype(F32) for n in names}
    gg, gc = gpu.backward(diffs), cpu.backward(diffs)
    berr = rel_l2(gg, gc)
    report('fp32 layer %s K%d M%d %dpx' % (name, K, M, edge), {'forward_rel_l2': errs, 'relu_flips': flips, 'backward_rel_l2': berr})
    assert berr <= 3 * tol, berr
    # the same layer on the split-operand Winograd kernel (st_set_conv_algo 2: six bf16 partial products per transform-domain product,
    # conv3x3_wino_split.hip), at the SAME bars, against the same oracle evaluation
    del gg, gc
    gpu.engine.set_conv_algo(2)
    fs = gpu.forward(x, names)
    errs2 = {n: rel_l2(fs[n], fc[n]) for n in names}
    flips2 = int(np.sum((fs['conv_b'] > 0) != (fc['conv_b'] > 0)))
    assert not np.array_equal(fs['conv_b'], fg['conv_b'])         # the other kernel ran
    for n in names:
        assert errs2[n] <= tol, (n, errs2)
    assert flips2 <= 1e-5 * fc['conv_b'].size + 5, flips2
    cpu.adopt_forward_state(fs)
    berr2 = rel_l2(gpu.backward(diffs), cpu.backward(diffs))
    report('fp32 layer %s K%d M%d %dpx, split-operand Winograd' % (name, K, M, edge), {'forward_rel_l2': errs2, 'relu_flips': flips2, 'backward_rel_l2': berr2})
    assert berr2 <= 3 * tol, berr2


@pytest.mark.parametrize('name,K,M,edge,pooled', VGG_SHAPES)
def test_production_conv_shapes_bf16_at_2048(name, K, M, edge, pooled):
    """Rounded-operand oracle fed with the GPU's own conv_a blob (isolates conv_b from upstream rounding)."""
    edge *= 2
    # the two widest layers on the production ROW length but half the rows (thousands of workgroups either way: the same tiles and
    # pipelines are chosen; the numpy side of the square case cost 32 s of box time)
    topo, params, gpu, cpu, x = _layer_case(K, M, edge, pooled, 'bf16', rows=edge // 2 if K <= 128 and M <= 128 and edge >= 1024 else None)
    names = [l[1] for l in topo]
    fg = gpu.forward(x, names)
    wgt, b = params['conv_b']
    ref = np.maximum(conv3x3_forward(bf16_round(fg['conv_a'][0]), bf16_round(wgt), b), 0)
    ferr = rel_l2(fg['conv_b'][0], ref)
    assert ferr <= 2e-6, ferr                   # measured 1.3e-7 .. 3.6e-7 on MI355X
    if pooled:
        assert np.array_equal(fg['pool_b'][0], maxpool_forward(fg['conv_b'][0])[0])      # pooling is exact
    del ref
    cpu.forward(x, names)
    cpu.adopt_forward_state(fg)
    rng = np.random.RandomState(7)
    diffs = {n: rng.randn(*fg[n].shape).astype(F32) for n in names}
    berr = rel_l2(gpu.backward(diffs), cpu.backward(diffs))
    report('bf16 layer %s K%d M%d %dpx' % (name, K, M, edge), {'forward_rel_l2': ferr, 'backward_rel_l2': berr})
    # the chain ends in conv_a's data gradient, whose diff operand is rounded to bf16 on both sides: the 1e-6 difference of
    # conv_b's output moves ~2.5e-4 of the elements across a rounding boundary (2^-9 each) -> ~2e-5 (4.7e-7 .. 1.3e-6 before
    # the first layer's dgrad took bf16 operands)
    assert berr <= 5e-5, berr


# ------------------------------------------------------------------------------ 2 + 3. whole objective, configs[1]
def _jobs(size, precision):
    content, style, init = images(size)
    topo = oracle.VGG19_TOPOLOGY
    params = oracle.he_init_weights(topo, seed=0)
    net = oracle.NetOracle(topo, params, full_forward=False, operands='bf16' if precision == 'bf16' else 'fp32')
    cpu = oracle.TransferOracle(net)
    cpu.feature_layers = WEIGHTED                # (content features / style Grams of the weighted blobs only: same values, a third of the set-up time)
    dev = st2.StyleTransfer(st2.HipModel(params, precision=precision))
    for st in (cpu, dev):
        st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
        st.set_weights(WEIGHTS, PARAMS)
    return net, cpu, dev


def _flip_positions(net, eng, names, blobs=None):
    """ReLU-sign flips over the conv blobs and arg-max flips over the pools between the two forwards just run: per blob name the
    (n, 2) array of (y, x) positions (any channel) where the two implementations took different branches.  `blobs`: the GPU's
    blobs if the caller already fetched them ({name: (1, C, h, w)})."""
    out, relu, pool, total = {}, 0, 0, 0
    get = (lambda n: blobs[n][0]) if blobs is not None else (lambda n: eng.get_blob(n)[0])
    for i, n in enumerate(names):
        c = net._blobs[n]
        if n.startswith('conv'):
            diff = (get(n) > 0) != (c > 0)
            relu += int(diff.sum())
            total += c.size
        else:
            diff = maxpool_forward(get(names[i - 1]))[1] != net._slots[n]
            pool += int(diff.sum())
        out[n] = np.argwhere(diff.any(axis=0))
    return out, relu, pool, total


def _record_backward(net):
    """Keep what the oracle's objective hands to its ranged backward (worker.py:295: model.backward(diffs)) and what came back, so
    that the SAME injected diffs can be back-propagated again on another forward state."""
    rec, orig = {}, net.backward

    def backward(diffs):
        rec['diffs'] = dict(diffs)
        out = orig(diffs)
        rec['scd'] = out.copy()
        return out
    net.backward = backward
    return rec, orig


STYLE_LAYERS = [n for n in WEIGHTS['style']]


@pytest.fixture(scope='module')
def fp32_1024():
    """configs[1] on both sides, evaluated ONCE for every test of this module that needs it (an oracle evaluation at this size
    is ~15 s of CPU time): the two jobs, the first objective evaluation, where the two forwards took different branches, and --
    for the two assertions that replace round 3's explanations -- the oracle's gradient re-derived (backward only) with
    (a) the GPU's branch decisions adopted, (b) additionally the engine's own D = G - G_style in the style terms."""
    net, cpu, dev = _jobs(1024, 'fp32')
    rec, plain_backward = _record_backward(net)
    lo, go = cpu.opfunc(cpu.input)
    net.backward = plain_backward
    ld, gd = dev.opfunc()
    eng = dev.engine
    names = ['data'] + TO_CONV5_1
    blobs = {n: eng.get_blob(n) for n in names}
    ferr = {n: rel_l2(blobs[n][0], net._blobs[n]) for n in WEIGHTED}
    flips, relu, pool, total = _flip_positions(net, eng, TO_CONV5_1, blobs)
    # (b) the engine's D of every style layer: Gram of the current features (the forward dev.opfunc just ran) minus the Gram of the
    # style image's features, both by the engine's own Gram kernels (st_gram: the reduction the objective uses)
    g_cur = {n: eng.gram(n) for n in STYLE_LAYERS}
    aux = st2.HipModel(net.params)
    aux.forward(net.preprocess(images(1024)[1]), ['conv5_1'])
    g_sty = {n: aux.engine.gram(n) for n in STYLE_LAYERS}
    del aux
    diffs_b, d_rel = dict(rec['diffs']), {}
    for n in STYLE_LAYERS:
        feat = net._blobs[n]                                   # the ORACLE's features (still those of its evaluation at x0)
        c, hw = feat.shape[0], feat.shape[1] * feat.shape[2]
        f2 = feat.reshape(c, hw)
        d_o = oracle.gram(feat[None]) - cpu.grams[n]
        d_e = (g_cur[n] - g_sty[n]).astype(F32)
        # everything downstream of D in the oracle's own arithmetic (worker.py:262-269): S = c2 D F, the first-evaluation norm
        # N_s = rms(S) -- a function of D too -- and the injected (sw / N_s) S
        c2, sw = F32(2.0 / (d_o.size * f2.size)), F32(WEIGHTS['style'][n])
        s_o, s_e = c2 * np.dot(d_o, f2), c2 * np.dot(d_e, f2)
        n_o, n_e = cpu.norms['s'][n], np.sqrt(np.mean(s_e ** 2))
        diffs_b[n] = rec['diffs'][n] + ((sw / n_e) * s_e - (sw / n_o) * s_o).reshape(rec['diffs'][n].shape).astype(F32)
        d_rel[n] = {'D_rel_l2_engine_vs_oracle': rel_l2(d_e, d_o), 'G_over_D': float(np.linalg.norm(g_cur[n]) / np.linalg.norm(d_e)),
                    'norm_rel': float(abs(n_e - n_o) / n_o)}
    # the same evaluation with the other conv algorithms (2: split-operand Winograd; 0: direct kernel only, measured once for the record
    # when ST2_CENSUS_DIRECT=1): forward errors, branch flips against the oracle's forward (still the oracle's own state here), gradient
    others = {}
    for algo in (2, 0) if os.environ.get('ST2_CENSUS_DIRECT') == '1' else (2,):
        job = st2.StyleTransfer(st2.HipModel(net.params))
        job.engine.set_conv_algo(algo)
        content, style, init = images(1024)
        job.set_input(init); job.set_content(content); job.set_style(style); job.reset()
        job.set_weights(WEIGHTS, PARAMS)
        l2, g2 = job.opfunc()
        b2 = {n: job.engine.get_blob(n) for n in names}
        f2, relu2, pool2, _ = _flip_positions(net, job.engine, TO_CONV5_1, b2)
        others[algo] = dict(ld=l2, gd=g2, blobs=b2, relu=relu2, pool=pool2, ferr={n: rel_l2(b2[n][0], net._blobs[n]) for n in WEIGHTED},
                            conv_ferr={n: rel_l2(b2[n][0], net._blobs[n]) for n in TO_CONV5_1 if n.startswith('conv')})
        del job
    # (a) the oracle's real backward on the diffs of ITS evaluation, masks / arg-max taken from the GPU's forward
    net.adopt_forward_state(blobs)
    go_adopt = go - rec['scd'] + net.backward(rec['diffs'])
    go_adopt_d = go - rec['scd'] + net.backward(diffs_b)
    del blobs
    for algo, o in others.items():
        net.adopt_forward_state(o.pop('blobs'))
        o['go_adopt'] = go - rec['scd'] + net.backward(rec['diffs'])
    return dict(net=net, cpu=cpu, dev=dev, lo=lo, go=go, ld=ld, gd=gd, ferr=ferr, flips=flips, relu=relu, pool=pool, total=total,
                go_adopt=go_adopt, go_adopt_d=go_adopt_d, d_rel=d_rel, others=others,
                tc=dict(cpu.traces[-1].data), td=dict(dev.traces[-1].data), x0=cpu.input.copy())


def test_vgg19_objective_fp32_at_1024_bench_inputs(fp32_1024):
    s = fp32_1024
    lo, go, ld, gd, ferr = s['lo'], s['go'], s['ld'], s['gd'], s['ferr']
    # (2a) forward blobs of the six weighted layers
    for n in WEIGHTED:
        assert ferr[n] <= 3e-6, ferr               # measured 1.2e-7 (conv1_1) .. 9.5e-7 (conv5_1)
    # (2b) loss and every trace scalar that does not depend on the backward pass
    assert np.isclose(ld, lo, rtol=1e-5), (ld, lo)
    tc, td = s['tc'], s['td']
    assert list(td) == list(tc)
    for k in tc:
        if k.endswith('_loss') or k.endswith('_c_grad') or k.endswith('_s_grad') or k in ('t_grad', 'p_grad'):
            assert np.isclose(td[k], tc[k], rtol=2e-4), (k, td[k], tc[k])
    # (3) end-to-end gradient, flips counted
    relu, pool, total = s['relu'], s['pool'], s['total']
    gerr = rel_l2(gd, go)
    pix = np.abs(gd - go)[0].max(0)
    frac = float(np.mean(pix > 1e-3 * np.abs(go).max()))
    report('fp32 vgg19 1024 objective', {'forward_rel_l2': ferr, 'loss_rel': float(abs(ld - lo) / abs(lo)), 'grad_rel_l2': gerr,
                                         'relu_flips': relu, 'pool_argmax_flips': pool, 'activations': total,
                                         'affected_pixel_frac': frac, 'trace_grad_rms': [td['grad'], tc['grad']]})
    # measured on MI355X: 46 ReLU + 20 arg-max flips in 3.04e8 activations, gradient rel-L2 6.9e-4, 0.66 % of the pixels
    assert relu <= 1e-6 * total and pool <= 100, (relu, pool, total)
    assert gerr <= 3e-3, gerr
    assert frac <= 0.02, frac
    assert np.isclose(td['grad'], tc['grad'], rtol=1e-3) and np.isclose(td['scd_grad'], tc['scd_grad'], rtol=1e-3)


def test_the_1024_gradient_differs_only_inside_the_receptive_fields_of_flipped_activations(fp32_1024):
    """BASELINE.md section 3 asks for a per-step gradient rel-L2 <= 1e-4 on the fp32 path.  ReLU and max-pool are discontinuous:
    two correct fp32 forwards (3e-7 .. 9e-7 apart) take different branches at a few dozen of 3e8 activations, and each such flip
    changes the gradient by O(1) -- but ONLY inside the image-space receptive field of the flipped unit (3 px for conv1_1 ...
    156 px for conv5_1).  So: paint those receptive fields, and hold everything OUTSIDE them to the gate."""
    s = fp32_1024
    go, gd = s['go'][0], s['gd'][0]
    geo = receptive_geometry(oracle.VGG19_TOPOLOGY[:17])
    names = ['data'] + TO_CONV5_1
    mask = np.zeros(go.shape[1:], bool)
    per_layer = {}
    for n, pos in s['flips'].items():
        if len(pos):
            paint_receptive_fields(mask, pos, geo[names.index(n)])
            per_layer[n] = int(len(pos))
    frac = float(mask.mean())
    out = ~mask
    err_out = float(np.linalg.norm((gd - go)[:, out].astype(np.float64)) / np.linalg.norm(go[:, out].astype(np.float64)))
    err_in = float(np.linalg.norm((gd - go)[:, mask].astype(np.float64)) / max(np.linalg.norm(go[:, mask].astype(np.float64)), 1e-30)) if mask.any() else 0.0
    worst_out = float(np.abs(gd - go)[:, out].max() / np.abs(go).max())
    report('fp32 vgg19 1024 gradient, flip-attributed', {'flipped_positions_per_blob': per_layer, 'receptive_field_union_frac': frac,
                                                        'rel_l2_outside': err_out, 'rel_l2_inside': err_in, 'max_abs_outside_over_max_grad': worst_out,
                                                        'rel_l2_everywhere': rel_l2(gd, go)})
    # measured on MI355X: 66 flipped positions over 17 blobs, their receptive fields cover 18.6 % of the image (a single conv4_4 /
    # conv5_1 flip paints 1.3 % / 2.3 % of a 1024^2 image); rel-L2 outside 1.9e-5, inside 1.5e-3, everywhere 6.5e-4.  What is left
    # outside is not the backward arithmetic (8.8e-7 on a shared forward state, test below) but the fp32 summation order of the
    # million-term Gram sums entering D = G - G_style on both sides.
    assert err_out <= 1e-4, (err_out, frac)        # BASELINE.md section 3's gate, met where it can be
    assert err_out <= 4e-5, err_out                # (and with margin)
    assert worst_out <= 1e-4, worst_out
    assert frac <= 0.35, frac
    if s['relu'] + s['pool']:
        assert err_in >= 10 * err_out              # the disagreement sits inside


def test_with_the_gpus_branch_decisions_adopted_the_1024_gradient_agrees_everywhere(fp32_1024):
    """Round 3 EXPLAINED the 1.5e-3 inside the painted receptive fields as "only the 66 flips"; this asserts it.  The oracle's real
    objective backward (worker.py:88-106 through oracle.NetOracle.backward, on the diffs its own opfunc injected, worker.py:242-277)
    is run again with the ReLU masks and pool arg-max of the GPU's forward (NetOracle.adopt_forward_state): nothing else changes.
    If flips are the whole story the gradient then agrees EVERYWHERE as well as it did outside the painted fields (1.9e-5).
    Bar stated before the first run: 2.5e-5 (VERDICT r3 asked for 2e-5; 1.9e-5 was the outside figure, the bar leaves 30 %).
    Measured on MI355X: 6.52e-4 plain -> 1.905e-5 with the 66 branch decisions adopted; largest single deviation 2.6e-5 of the
    largest gradient."""
    s = fp32_1024
    go, gd, ga = s['go'], s['gd'], s['go_adopt']
    err_plain, err_adopt = rel_l2(gd, go), rel_l2(gd, ga)
    worst = float(np.abs(gd - ga).max() / np.abs(ga).max())
    report('fp32 vgg19 1024 gradient, GPU branch decisions adopted by the oracle', {
        'rel_l2_everywhere_plain': err_plain, 'rel_l2_everywhere_adopted': err_adopt, 'max_abs_over_max_grad_adopted': worst,
        'flips': s['relu'] + s['pool']})
    assert err_adopt <= 2.5e-5, (err_adopt, err_plain)
    assert worst <= 1e-4, worst
    if s['relu'] + s['pool']:
        assert err_adopt <= 0.2 * err_plain, (err_adopt, err_plain)      # the flips were (at least) 80 % of the disagreement


def test_the_split_operand_convs_meet_the_same_bars_at_1024(fp32_1024):
    """configs[1]'s objective with every eligible conv on the split-operand Winograd kernel (st_set_conv_algo 2), against the SAME oracle
    evaluation as the IEEE-fp32 path above and at its bars (stated before the first run): forward blobs of the
    weighted layers <= 3e-6, loss rtol 1e-5, branch flips not above the fp32 Winograd path's (66 measured in round 4; bar 80), gradient
    with the GPU's branch decisions adopted by the oracle <= 2.5e-5 at every pixel.  With ST2_CENSUS_DIRECT=1 the direct kernel
    (st_set_conv_algo 0, the im2col-free implicit GEMM `north_star` names) is measured the same way, for the record: how many of the
    flips are Winograd's."""
    s = fp32_1024
    lo, go = s['lo'], s['go']
    rows = {1: {'relu_flips': s['relu'], 'pool_argmax_flips': s['pool'], 'grad_rel_l2': rel_l2(s['gd'], go),
                'grad_rel_l2_branch_decisions_adopted': rel_l2(s['gd'], s['go_adopt']), 'forward_rel_l2': s['ferr'],
                'loss_rel': float(abs(s['ld'] - lo) / abs(lo))}}
    for algo, o in s['others'].items():
        rows[algo] = {'relu_flips': o['relu'], 'pool_argmax_flips': o['pool'], 'grad_rel_l2': rel_l2(o['gd'], go),
                      'grad_rel_l2_branch_decisions_adopted': rel_l2(o['gd'], o['go_adopt']), 'forward_rel_l2': o['ferr'],
                      'forward_rel_l2_every_conv_blob': o['conv_ferr'], 'loss_rel': float(abs(o['ld'] - lo) / abs(lo))}
    report('fp32 vgg19 1024 objective by conv algorithm (0 direct, 1 Winograd fp32, 2 split-operand Winograd)', {str(k): v for k, v in sorted(rows.items())})
    o = s['others'][2]
    for n in WEIGHTED:
        assert o['ferr'][n] <= 3e-6, o['ferr']
    assert np.isclose(o['ld'], lo, rtol=1e-5), (o['ld'], lo)
    assert o['relu'] + o['pool'] <= 80, (o['relu'], o['pool'])
    assert rows[2]['grad_rel_l2_branch_decisions_adopted'] <= 2.5e-5, rows[2]
    assert rows[2]['grad_rel_l2'] <= 3e-3


def test_the_residual_after_adoption_is_the_gram_difference(fp32_1024):
    """Round 3 ATTRIBUTED the 1.9e-5 that remains to "the fp32 summation order of the million-term Gram sums that enter
    D = G - G_style" (content and style are both noise images here: G is nearly G_style, D a small difference of large sums).
    Asserted: hand the oracle the ENGINE's D in its style terms (same features, same norms, same backward on the adopted state)
    and the residual must fall below 5e-6 -- a quarter of what it was -- or that sentence is wrong.
    Measured on MI355X: 1.905e-5 -> 6.6e-7.  |G| / |D| is 277 (conv1_1) .. 576 (conv3_1) on these inputs, the two D differ by
    0.9e-4 .. 2.8e-4 in relative L2, the first-evaluation norms N_s by up to 2.6e-5."""
    s = fp32_1024
    gd = s['gd']
    err_adopt, err_d = rel_l2(gd, s['go_adopt']), rel_l2(gd, s['go_adopt_d'])
    report('fp32 vgg19 1024 gradient, adopted branches + the engine\'s D in the style terms', {
        'rel_l2_adopted': err_adopt, 'rel_l2_adopted_with_engine_D': err_d, 'per_style_layer': s['d_rel']})
    assert err_d <= 5e-6, (err_d, err_adopt)
    assert err_d <= 0.5 * err_adopt, (err_d, err_adopt)


def test_adam_steps_with_frozen_norms_at_1024(fp32_1024):
    """worker.py:303-310 + optimizers.py:20-27 at the headline size: two Adam iterations from the initial image on both sides (the
    first captures the norms, the second evaluates with them frozen; a third was run until round 4 and cost 9 s of oracle time) -- per-step loss and the ITERATE itself (north_star: "output
    pixels match the reference CPU worker ... within a stated fp32 MSE tolerance")."""
    s = fp32_1024
    cpu, dev, go = s['cpu'], s['dev'], s['go']
    # (the second evaluation with frozen norms at a moved image is covered at this size by the Adam steps below -- every step after the
    #  first evaluates with frozen norms -- and at 96 x 128 / in the golden vectors by the dedicated tests; evaluating it separately
    #  here cost 7 s of oracle time)
    # three Adam iterations (SetOptimizer resets the state and the norms on both sides, worker.py:387-391,172-175)
    cpu.input[:] = s['x0']
    dev.engine.set_input_nchw(s['x0'])
    cpu.set_optimizer('adam', 10)
    dev.optimizer_cls = st2.AdamOptimizer; dev.set_step_size(10); dev.reset()
    assert cpu.start() and dev.start()
    steps = []
    for i in range(2):
        ic, tc = cpu.step()
        idv, td = dev.step()
        assert list(td) == list(tc)
        mse = float(np.mean((idv.astype(np.float64) - ic) ** 2))
        steps.append({'loss_rel': float(abs(td['loss'] - tc['loss']) / abs(tc['loss'])), 'image_mse': mse,
                      'image_max_abs': float(np.max(np.abs(idv - ic))), 'pixels_off_by_more_than_1': float(np.mean(np.abs(idv - ic) > 1.0))})
        assert np.isclose(td['loss'], tc['loss'], rtol=1e-4), (i, td['loss'], tc['loss'])
    report('fp32 vgg19 1024 adam steps', {'steps': steps, 'moved_mse': float(np.mean((ic - s['x0'][0].transpose(1, 2, 0) - net_mean()) ** 2))})
    # Adam's first step is sign-like (x -= 10 m^ / sqrt(v^) = 10 sign(g)): a pixel whose tiny gradient has the other sign on the
    # other side lands 20 levels away, everything else agrees to rounding.  Stated tolerance, 0..255 units:
    assert steps[-1]['image_mse'] <= 0.5, steps                             # measured 0.009 / 0.14 (/ 0.19) after 1 / 2 (/ 3) steps (the image moved by MSE 79 in three)
    assert steps[-1]['pixels_off_by_more_than_1'] <= 1e-2, steps            # measured 3.3e-3 after three steps, 2e-5 after the first


def net_mean():
    return oracle.NetOracle.mean.transpose(1, 2, 0)


def test_vgg19_ranged_backward_fp32_at_1024_on_shared_forward_state():
    """worker.py:88-106 at full size: injections at conv, pool and data blobs, masks / arg-max taken from the GPU's
    own forward (so that only the backward arithmetic is compared)."""
    topo = oracle.VGG19_TOPOLOGY
    params = oracle.he_init_weights(topo, seed=0)
    net = oracle.NetOracle(topo, params, full_forward=False)
    gpu = st2.HipModel(params)
    x = net.preprocess(images(1024)[2])
    full = gpu.forward(x, ['data'] + TO_CONV5_1)
    net.forward(x, ['conv5_1'])
    net.adopt_forward_state(full)
    rs = np.random.RandomState
    diffs = {n: rs(5 + i).randn(*full[n].shape).astype(F32) for i, n in enumerate(['conv5_1', 'pool4', 'conv4_2', 'conv3_1', 'conv2_1', 'conv1_1', 'data'])}
    err = rel_l2(gpu.backward(diffs), net.backward(diffs))
    report('fp32 vgg19 1024 ranged backward (shared forward state)', {'rel_l2': err})
    assert err <= 3e-6, err                     # measured 8.8e-7


# ------------------------------------------------------------------------------ configs[2]: 2048^2, bf16, L-BFGS
def test_vgg19_objective_bf16_at_2048():
    """Whole objective against the rounded-operand oracle.  Two correct bf16 implementations decorrelate to the bf16
    noise floor within a few layers (a 1e-6 fp32 difference moves ~2.5e-4 of a blob's elements across a bf16 rounding
    boundary, tests/test_gpu_bf16.py), hence the loss / cosine bars; the per-layer arithmetic is pinned tightly by
    test_production_conv_shapes_bf16_at_2048."""
    net, cpu, dev = _jobs(2048, 'bf16')
    cpu.set_optimizer('lbfgs', 1)
    dev.optimizer_cls = st2.LBFGSOptimizer; dev.set_step_size(1); dev.reset()
    assert cpu.start() and dev.start()
    lo, go = cpu.opfunc(cpu.input)
    ld, gd = dev.opfunc()
    eng = dev.engine
    # the lean flow keeps style-only blobs as bf16 copies: take the fp32 blobs from the bit-identical full flow of the same input
    eng.set_precision('bf16-full')
    dev.opfunc()
    ferr = {n: rel_l2(eng.get_blob(n)[0], net._blobs[n]) for n in WEIGHTED}
    eng.set_precision('bf16')
    cos = float(np.vdot(gd.astype(np.float64), go.astype(np.float64)) / (np.linalg.norm(gd.astype(np.float64)) * np.linalg.norm(go.astype(np.float64))))
    vals = {'forward_rel_l2': ferr, 'loss_rel': float(abs(ld - lo) / abs(lo)), 'grad_rel_l2': rel_l2(gd, go), 'grad_cosine': cos}
    report('bf16 vgg19 2048 objective', vals)
    # measured: forward 1.2e-7 (conv1_1, fp32 operands) .. 4.4e-3 (conv5_1), loss 2.4e-4, gradient rel-L2 3.3e-2, cosine 0.99947
    assert ferr['conv1_1'] <= 1e-6 and max(ferr.values()) <= 2e-2, ferr
    assert np.isclose(ld, lo, rtol=3e-3)
    assert cos >= 0.998 and vals['grad_rel_l2'] <= 8e-2
    # (Round 3 also took ONE L-BFGS step here against the oracle -- a second 30 s oracle evaluation at this size for a step whose
    #  direction is the gradient just compared, normalised to unit RMS (optimizers.py:97-99).  The optimizer at this size is now checked
    #  where it says more: bf16 against the fp32 engine over five L-BFGS steps at 1536 x 2048 and twenty at 768 x 1024 on a contracting
    #  workload, and the engine's bf16 drift against the rounded-operand oracle's own over eleven steps, below.)


# ------------------------------------------------------------------------------ bf16 feature path against the fp32 engine, over many steps
def _engine_job(size, precision, params=None, optimizer='lbfgs'):
    content, style, init = images(size)
    params = params if params is not None else oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0)
    job = st2.StyleTransfer(st2.HipModel(params, precision=precision))
    job.set_input(init); job.set_content(content); job.set_style(style); job.reset()
    job.set_weights(WEIGHTS, PARAMS)
    job.optimizer_cls = {'adam': st2.AdamOptimizer, 'lbfgs': st2.LBFGSOptimizer}[optimizer]
    job.set_step_size({'adam': 10, 'lbfgs': 1}[optimizer])
    job.reset()
    assert job.start()
    return job


@pytest.mark.parametrize('size,optimizer,steps,stable,rtol', [(1024, 'adam', 20, 20, 0.1), (1024, 'lbfgs', 6, 4, 2e-2), (2048, 'lbfgs', 5, 3, 2e-2)])
def test_bf16_follows_the_fp32_loss_curve_at_size(size, optimizer, steps, stable, rtol):
    """configs[2] ("bf16 features / fp32 Gram") against the fp32 engine (itself checked against the oracle above) on the same job,
    step by step: the loss curve and the final iterate.  A single objective evaluation differs by the bf16 rounding of every conv
    operand (gradient cosine 0.9995); what matters to a user is that the OPTIMISATION goes the same way.
    The reference's L-BFGS takes fixed steps without a line search (optimizers.py:62-77): on this workload (uniform-noise images,
    step size 1) the fp32 loss itself turns around after the fourth step and then wanders (measured: 2.0e9, 3.0e7, 2.9e7, 2.8e7,
    2.9e7, 3.6e7, 4.8e7 ... 4.2e8), so two runs can be compared only while the iteration is still contracting: the first `stable`
    steps.  Adam (the headline optimizer) is compared over all 20 steps: with the reference's step size 10 on noise images its loss
    swings between 2e9 and 1e11 from step to step; the bf16 run follows every swing within 0.002 % .. 6.7 % (measured)."""
    a, b = _engine_job(size, 'fp32', optimizer=optimizer), _engine_job(size, 'bf16', optimizer=optimizer)
    curve = []
    for i in range(steps):
        ia, ta = a.step()
        ib, tb = b.step()
        curve.append((ta['loss'], tb['loss']))
        if i == stable - 1:
            mse = float(np.mean((ia.astype(np.float64) - ib) ** 2))
            moved = float(np.mean((ia.astype(np.float64) - images(size)[2]) ** 2))
    rel = [abs(y - x) / abs(x) for x, y in curve]
    report('bf16 vs fp32 engine, %d %s steps at %d' % (steps, optimizer, size),
           {'loss_fp32': [c[0] for c in curve], 'loss_bf16': [c[1] for c in curve], 'loss_rel': rel, 'compared_steps': stable,
            'image_mse_after_compared_steps': mse, 'fp32_moved_mse': moved})
    if optimizer == 'lbfgs':
        assert curve[stable - 1][0] < curve[0][0] and curve[stable - 1][1] < curve[0][1]      # both descend
    # (Adam at step size 10 does not: its first step throws the loss from 3.8e7 to 1.4e11 -- in the reference too, the oracle's
    # three-step test above agrees with the engine to 7e-5 on exactly that curve)
    assert max(rel[:stable]) <= rtol, rel                                    # the two loss curves stay together while the iteration is stable
    if optimizer == 'lbfgs':
        assert mse <= 0.1 * moved + 0.05, (mse, moved)                        # the iterates differ by a small part of how far they moved
    # (Adam at step size 10 moves every pixel by +-10 per step, sign-like: the bf16 noise on small gradient components flips signs and
    # the two IMAGES random-walk apart -- measured MSE 321 after 20 steps against a move of 486 -- while the loss curves stay together)


# ------------------------------------------------------------------------------ an image-like job at size (a workload on which L-BFGS contracts)
def _image_like(fit):
    """The reference's example pair (decoded pixels: tests/golden/config1_sources.npz) fitted to `fit` px as app.py would
    (utils.resize_to_fit: content 768 x 1024 / style 640 x 1024 at fit = 1024), and an ITERATE-like initial image: the content
    image with +-16 levels of seeded noise -- what app.py:259 re-sends as `input_image` after a worker respawn is the last iterate,
    an image near the content image.  (The content image itself cannot be the input: x == content makes N_c = 0 and the reference's
    objective NaN, worker.py:253-256 -- test_input_equal_to_content_gives_the_references_nan.)  On these inputs the reference's
    fixed-step L-BFGS contracts (CPU oracle at 192 x 256: loss 1.27e8 -> 2.1e6 over 20 steps, monotone but for three steps), unlike
    on the uniform-noise bench inputs, so whole trajectories can be compared."""
    from PIL import Image
    from style_transfer2_amd import jobs
    from helpers import load
    src = load('config1_sources.npz')
    content = np.uint8(jobs.resize_to_fit(Image.fromarray(src['golden_gate']), fit))
    style = np.uint8(jobs.resize_to_fit(Image.fromarray(src['starry_night']), fit))
    init = np.clip(content.astype(np.int32) + np.random.RandomState(5).randint(-16, 17, content.shape), 0, 255).astype(np.uint8)
    return content, style, init


def _image_like_engine_run(inputs, precision, steps, keep):
    content, style, init = inputs
    job = st2.StyleTransfer(st2.HipModel(oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0), precision=precision))
    job.set_input(init); job.set_content(content); job.set_style(style); job.reset()
    job.set_weights(WEIGHTS, PARAMS)
    job.optimizer_cls = st2.LBFGSOptimizer; job.set_step_size(1); job.reset()          # the reference's default optimizer (worker.py:135-136)
    assert job.start()
    losses, kept = [], {}
    for i in range(steps):
        img, tr = job.step()
        losses.append(tr['loss'])
        if i + 1 in keep:
            kept[i + 1] = img.copy()
    return losses, kept


@pytest.fixture(scope='module')
def image_like_1024():
    inputs = _image_like(1024)
    losses, kept = _image_like_engine_run(inputs, 'fp32', 20, (5, 20))
    return dict(inputs=inputs, losses=losses, images=kept)


def _trajectory_report(key, la, lb, ia, ib, init):
    rel = [abs(y - x) / abs(x) for x, y in zip(la, lb)]
    mse = float(np.mean((ia.astype(np.float64) - ib) ** 2))
    moved = float(np.mean((ia.astype(np.float64) - init) ** 2))
    report(key, {'loss_a': [float(v) for v in la], 'loss_b': [float(v) for v in lb], 'loss_rel': rel, 'final_image_mse': mse, 'moved_mse': moved})
    return rel, mse, moved


def test_image_like_job_fp32_engine_follows_the_oracle_over_five_lbfgs_steps(image_like_1024):
    """optimizers.py:62-108 + worker.py:231-310 on an image-like job at size (768 x 1024), the fp32 engine against the CPU oracle,
    five L-BFGS steps (six objective evaluations).  Bars stated before the first run: per-step loss rtol
    1e-4, final iterate within 5 % (MSE) of how far it moved.  The oracle's side is stored since round 5 (tests/golden/make_trajectories.py
    `size`: its five losses and every fourth pixel of its fifth iterate -- the MSE is taken over that sample; the CPU suite re-runs the
    first step): 75 s of box time less."""
    from helpers import load
    s = image_like_1024
    content, style, init = s['inputs']
    g = load('oracle_trajectories.npz')
    lc, ic = list(g['size_losses_fp32']), g['size_final_fp32_sub4']
    rel, mse, moved = _trajectory_report('image-like 768x1024, 5 L-BFGS steps: fp32 engine vs oracle', lc, s['losses'][:5], ic,
                                         s['images'][5][::4, ::4, :], init[::4, ::4, :])
    assert lc[-1] < 0.5 * lc[0], lc                               # the iteration contracts on this workload
    assert max(rel) <= 1e-4, rel
    assert mse <= 0.05 * moved, (mse, moved)


def test_image_like_job_bf16_engine_follows_the_rounded_operand_oracle_at_size(image_like_1024):
    """configs[2]'s optimizer against the ORACLE at size again (round 4 compared the bf16 engine at size with the fp32
    engine only): the same image-like job at 768 x 1024, bf16 conv operands, three L-BFGS steps (four objective evaluations) against the
    stored per-step losses of the rounded-operand oracle (NetOracle(operands='bf16'): every conv operand rounded to bf16, fp32
    accumulate; tests/golden/make_trajectories.py `size`).  Bars stated in the commit that added the fixture, before the first GPU run:
    3e-3 on the first step, 2e-2 after (the fixed-step iteration multiplies a difference by 4 - 10 per step; the two sides differ in
    summation order and in the Gram form of the recursion)."""
    from helpers import load
    lo = list(load('oracle_trajectories.npz')['size_losses_bf16'])
    lb, _ = _image_like_engine_run(image_like_1024['inputs'], 'bf16', 3, ())
    rel = [abs(b - a) / abs(a) for a, b in zip(lo, lb)]
    report('image-like 768x1024, 3 L-BFGS steps: bf16 engine vs rounded-operand oracle', {'loss_oracle': lo, 'loss_engine': [float(v) for v in lb], 'loss_rel': rel})
    assert rel[0] <= 3e-3, rel
    assert max(rel) <= 2e-2, rel


# What "bf16 follows fp32" can mean is set by the arithmetic, not by the kernels: the ROUNDED-OPERAND ORACLE (the same numpy code with
# every conv operand rounded to bf16) run against its own fp32 self on this job at 192 x 256, 20 L-BFGS steps, drifts by up to 12 % in
# the per-step loss and ends 10.3 % (MSE) of the move away (measured on the CPU before these tests were written; the first of the
# tests below re-measures it at 12 steps next to the engine).  The bars first stated for the engine -- 1 % / 5 % -- were tighter than
# the reference arithmetic itself and failed at 2.6 % / 9.5 %; they are now anchored to that measurement.
BF16_LOSS_RTOL, BF16_MSE_OF_MOVE = 5e-2, 0.15


def test_bf16_engine_drifts_from_fp32_no_more_than_the_rounded_operand_oracle_does():
    """192 x 256 (the pair fitted to 256 px), 11 L-BFGS steps (through the roll-over at ten pairs), four runs: oracle fp32 / oracle
    with bf16 conv operands (both stored: tests/golden/oracle_trajectories.npz) / engine fp32 / engine bf16.  The engine's bf16-vs-fp32 drift (per-step loss, final iterate) must not
    exceed twice the oracle's own bf16-vs-fp32 drift."""
    inputs = _image_like(256)
    content, style, init = inputs
    # the two oracle trajectories are stored (tests/golden/make_trajectories.py made them with oracle.TransferOracle, 53 s of numpy;
    # tests/test_oracle_golden.py re-runs their first steps on the CPU): the box's time goes to the engine
    from helpers import load
    g = load('oracle_trajectories.npz')
    lo32, io32, lo16, io16 = list(g['drift_losses_fp32']), g['drift_final_fp32'], list(g['drift_losses_bf16']), g['drift_final_bf16']
    le32, ke32 = _image_like_engine_run(inputs, 'fp32', 11, (11,))
    le16, ke16 = _image_like_engine_run(inputs, 'bf16', 11, (11,))
    rel_o, mse_o, moved_o = _trajectory_report('image-like 192x256, 11 L-BFGS steps: ORACLE bf16 operands vs oracle fp32', lo32, lo16, io32, io16, init)
    rel_e, mse_e, moved_e = _trajectory_report('image-like 192x256, 11 L-BFGS steps: engine bf16 vs engine fp32', le32, le16, ke32[11], ke16[11], init)
    rel_x, mse_x, _ = _trajectory_report('image-like 192x256, 11 L-BFGS steps: engine fp32 vs oracle fp32', lo32, le32, io32, ke32[11], init)
    # fp32 against fp32 at this small size: a branch flip touches a larger share of the image and every L-BFGS step multiplies a
    # difference by ~4-10 (measured 6e-8, 7e-6, 1.5e-5, 5.6e-5, 7.6e-4 ...); the tight comparison at size is the five-step test above
    assert max(rel_x[:3]) <= 1e-4, rel_x
    assert max(rel_e) <= 2 * max(rel_o) + 1e-2, (rel_e, rel_o)
    assert mse_e / moved_e <= 2 * mse_o / moved_o + 0.02, (mse_e, moved_e, mse_o, moved_o)


def test_image_like_job_bf16_follows_the_fp32_engine_over_twenty_lbfgs_steps(image_like_1024):
    """configs[2]'s arithmetic (bf16 conv operands, fp32 accumulate / Gram / optimizer; Gram-form L-BFGS) against the fp32 engine
    on the same image-like job at 768 x 1024, twenty L-BFGS steps -- through the history roll-over at ten pairs.  Measured on MI355X:
    per-step loss within 2.6 %, final iterate 9.5 % (MSE) of the move away, final losses 1.274e8 / 1.240e8 from 2.56e9."""
    s = image_like_1024
    lb, kb = _image_like_engine_run(s['inputs'], 'bf16', 20, (20,))
    rel, mse, moved = _trajectory_report('image-like 768x1024, 20 L-BFGS steps: bf16 vs fp32 engine', s['losses'], lb, s['images'][20], kb[20], s['inputs'][2])
    assert s['losses'][-1] < 0.1 * s['losses'][0] and lb[-1] < 0.1 * lb[0], (s['losses'], lb)         # both contract by more than 10x
    assert max(rel) <= BF16_LOSS_RTOL, rel
    assert mse <= BF16_MSE_OF_MOVE * moved, (mse, moved)


def test_image_like_job_bf16_follows_the_fp32_engine_at_2048():
    """The same at configs[2]'s own size: the pair fitted to 2048 px (content 1536 x 2048), five L-BFGS steps, bf16 against fp32.
    Measured on MI355X: per-step loss within 3.0 %, final iterate 5.2 % (MSE) of the move away."""
    inputs = _image_like(2048)
    la, ka = _image_like_engine_run(inputs, 'fp32', 5, (5,))
    lb, kb = _image_like_engine_run(inputs, 'bf16', 5, (5,))
    rel, mse, moved = _trajectory_report('image-like 1536x2048, 5 L-BFGS steps: bf16 vs fp32 engine', la, lb, ka[5], kb[5], inputs[2])
    assert la[-1] < 0.5 * la[0] and lb[-1] < 0.5 * lb[0], (la, lb)
    assert max(rel) <= BF16_LOSS_RTOL, rel
    assert mse <= BF16_MSE_OF_MOVE * moved, (mse, moved)


# ------------------------------------------------------------------------------ weights with trained-like statistics
@pytest.mark.parametrize('name,K,M,edge,pooled,precision', [VGG_SHAPES[0] + ('fp32',), VGG_SHAPES[4] + ('fp32',), VGG_SHAPES[6] + ('fp32',),
                                                            VGG_SHAPES[4] + ('bf16',), VGG_SHAPES[6] + ('bf16',)])
def test_production_conv_shapes_with_trained_like_weights(name, K, M, edge, pooled, precision):
    """The production conv shapes again, with weights whose statistics are those of a trained network (per-channel gains over
    100x, non-zero biases, 5 % dead channels) instead of an initialisation: the Winograd / bf16 error bars are claims about
    arithmetic, not about He-normal weights."""
    edge *= 2 if precision == 'bf16' else 1
    topo = (('conv', 'conv_a', 3, K), ('conv', 'conv_b', K, M)) + ((('pool', 'pool_b'),) if pooled else ())
    params = trained_like_weights(topo, seed=K + M)
    gpu = st2.HipModel(params, topology=topo, precision=precision)
    cpu = oracle.NetOracle(topo, params, operands='bf16' if precision == 'bf16' else 'fp32')
    x = (np.random.RandomState(edge).randn(1, 3, edge, edge) * 40).astype(F32)
    names = [l[1] for l in topo]
    fg = gpu.forward(x, names)
    if precision == 'fp32':
        fc = cpu.forward(x, names)
        ferr = rel_l2(fg['conv_b'], fc['conv_b'])
        assert ferr <= (2e-6 if K <= 256 else 4e-6), ferr
        flips = int(np.sum((fg['conv_b'] > 0) != (fc['conv_b'] > 0)))
        assert flips <= 1e-5 * fc['conv_b'].size + 5, flips
    else:
        wgt, b = params['conv_b']
        ferr = rel_l2(fg['conv_b'][0], np.maximum(conv3x3_forward(bf16_round(fg['conv_a'][0]), bf16_round(wgt), b), 0))
        assert ferr <= 2e-6, ferr
        cpu.forward(x, names)
    dead = [c for c in range(M) if not params['conv_b'][0][c].any()]
    assert dead and not fg['conv_b'][0][dead].any()                        # dead channels are exactly zero after the in-place ReLU
    cpu.adopt_forward_state(fg)
    rng = np.random.RandomState(7)
    diffs = {n: rng.randn(*fg[n].shape).astype(F32) for n in names}
    berr = rel_l2(gpu.backward(diffs), cpu.backward(diffs))
    report('%s layer %s with trained-like weights' % (precision, name), {'forward_rel_l2': ferr, 'backward_rel_l2': berr, 'dead_channels': len(dead)})
    assert berr <= (1.2e-5 if precision == 'fp32' else 5e-5), berr


def test_vgg19_objective_with_trained_like_weights_at_96x128():
    """Whole VGG19 objective + 5 Adam steps, trained-like weights, against the oracle; the gradient is held to 1e-5 outside the
    receptive fields of whatever activations the two forwards flipped."""
    topo = oracle.VGG19_TOPOLOGY
    params = trained_like_weights(topo, seed=3)
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (96, 128, 3)).astype(np.uint8), rs(2).randint(0, 256, (80, 112, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (96, 128, 3)).astype(np.uint8))
    net = oracle.NetOracle(topo, params, full_forward=False)
    cpu, dev = oracle.TransferOracle(net), st2.StyleTransfer(st2.HipModel(params))
    for st in (cpu, dev):
        st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
        st.set_weights(WEIGHTS, PARAMS)
    lo, go = cpu.opfunc(cpu.input)
    ld, gd = dev.opfunc()
    assert np.isclose(ld, lo, rtol=2e-5), (ld, lo)
    check_trace(list(cpu.traces[-1].data), list(cpu.traces[-1].data.values()), {k: v for k, v in dev.traces[-1].data.items()}, rtol=1e-3,
                skip=('time', 'scd_grad', 'grad'))
    flips, relu, pool, total = _flip_positions(net, dev.engine, TO_CONV5_1)
    geo = receptive_geometry(topo[:17])
    names = ['data'] + TO_CONV5_1
    mask = np.zeros((96, 128), bool)
    for n, pos in flips.items():
        if len(pos):
            paint_receptive_fields(mask, pos, geo[names.index(n)])
    out = ~mask
    assert out.mean() >= 0.3, (relu, pool)
    err_out = float(np.linalg.norm((gd - go)[0][:, out].astype(np.float64)) / np.linalg.norm(go[0][:, out].astype(np.float64)))
    report('fp32 vgg19 96x128 trained-like weights', {'loss_rel': float(abs(ld - lo) / abs(lo)), 'relu_flips': relu, 'pool_flips': pool,
                                                      'grad_rel_l2_outside_flipped_fields': err_out, 'grad_rel_l2': rel_l2(gd, go)})
    assert err_out <= 1e-5, err_out
    cpu.set_optimizer('adam', 10)
    dev.optimizer_cls = st2.AdamOptimizer; dev.set_step_size(10); dev.reset()
    assert cpu.start() and dev.start()
    for i in range(5):
        ic, tc = cpu.step()
        idv, td = dev.step()
        assert np.isclose(td['loss'], tc['loss'], rtol=2e-4), (i, td['loss'], tc['loss'])
    assert np.mean((idv - ic) ** 2) <= 0.5


# ------------------------------------------------------------------------------ the reference's unguarded degenerate cases
def _tiny_pair(params, weights):
    topo = oracle.tiny_topology((8, 16), (2, 2))
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (24, 32, 3)).astype(np.uint8), rs(2).randint(0, 256, (20, 20, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (24, 32, 3)).astype(np.uint8))
    cpu, dev = oracle.TransferOracle(oracle.NetOracle(topo, params)), st2.StyleTransfer(st2.HipModel(params, topology=topo))
    for st in (cpu, dev):
        st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
        st.set_weights(weights, PARAMS)
    return cpu, dev, content


def _same_non_finite(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isposinf(a), np.isposinf(b)) and np.array_equal(np.isneginf(a), np.isneginf(b))


def test_input_equal_to_content_gives_the_references_nan():
    """worker.py:253-256 does not guard N_c = 0: with x == content the content gradient is identically zero, its RMS norm is 0, and
    loss and gradient become 0/0 = NaN.  The engine must not invent a guard the reference does not have: same NaNs."""
    topo = oracle.tiny_topology((8, 16), (2, 2))
    params = oracle.he_init_weights(topo, 0, 0.1)
    cpu, dev, content = _tiny_pair(params, {'content': {'conv2_2': 0.08}, 'style': {'conv1_1': 1}, 'deepdream': {}})
    cpu.set_input(content); dev.set_input(content)
    with np.errstate(all='ignore'):
        lo, go = cpu.opfunc(cpu.input)
    ld, gd = dev.opfunc()
    assert np.isnan(lo) and np.isnan(ld)
    assert np.isnan(go).all() and np.isnan(gd).all()
    tc, td = cpu.traces[-1].data, dev.traces[-1].data
    assert list(tc) == list(td)
    for k in tc:
        if k != 'time':
            assert _same_non_finite(tc[k], td[k]), (k, tc[k], td[k])
            if np.isfinite(tc[k]):
                assert np.isclose(td[k], tc[k], rtol=1e-3), (k, td[k], tc[k])


def test_a_dead_style_layer_gives_the_references_non_finite_values():
    """N_s = 0 (worker.py:264-266): a style layer whose features are identically zero (a dead layer: zero filters, negative bias)
    has S = 0, so its norm is 0; loss += sw * mean(D^2) / 0 and the injected diff is (sw / 0) * 0."""
    topo = oracle.tiny_topology((8, 16), (2, 2))
    params = oracle.he_init_weights(topo, 0, 0.1)
    w, b = params['conv1_2']
    params['conv1_2'] = (np.zeros_like(w), -np.ones_like(b))
    cpu, dev, _ = _tiny_pair(params, {'content': {'conv1_1': 0.08}, 'style': {'conv1_2': 1}, 'deepdream': {}})
    with np.errstate(all='ignore'):
        lo, go = cpu.opfunc(cpu.input)
    ld, gd = dev.opfunc()
    assert _same_non_finite(lo, ld) and not np.isfinite(lo)
    assert _same_non_finite(go, gd)
    tc, td = cpu.traces[-1].data, dev.traces[-1].data
    for k in tc:
        if k != 'time':
            assert _same_non_finite(tc[k], td[k]), (k, tc[k], td[k])
