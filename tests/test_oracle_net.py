"""Pins the oracle's network arithmetic (the part the reference delegates to Caffe, which cannot
run here): against torch CPU ops, against the independent C restatement, and against hand-computed
cases for the two Caffe rules a plain autograd VGG gets wrong (unmasked injection; first-max tie)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

import oracle
from oracle import caffe_net as cn, c_layers as cl

F32 = np.float32


@pytest.mark.parametrize('cin,cout,h,w', [(3, 8, 5, 7), (8, 16, 16, 20), (16, 8, 9, 13), (4, 4, 1, 1)])
def test_conv_and_pool_vs_torch_and_c(cin, cout, h, w):
    rng = np.random.RandomState(cin * 100 + h)
    x = rng.randn(cin, h, w).astype(F32)
    wt = rng.randn(cout, cin, 3, 3).astype(F32)
    b = rng.randn(cout).astype(F32)
    dy = rng.randn(cout, h, w).astype(F32)
    yt = Fn.conv2d(torch.tensor(x)[None], torch.tensor(wt), torch.tensor(b), padding=1)[0].numpy()
    dt = torch.nn.grad.conv2d_input((1, cin, h, w), torch.tensor(wt), torch.tensor(dy)[None],
                                    padding=1)[0].numpy()
    for impl in (cn, cl):
        assert np.allclose(impl.conv3x3_forward(x, wt, b), yt, rtol=1e-5, atol=1e-4)
        assert np.allclose(impl.conv3x3_backward_data(dy, wt), dt, rtol=1e-5, atol=1e-4)
    pt = Fn.max_pool2d(torch.tensor(x)[None], 2, 2, ceil_mode=True)[0].numpy()
    p1, slot = cn.maxpool_forward(x)
    p2, arg = cl.maxpool_forward(x)
    assert np.array_equal(p1, pt) and np.array_equal(p2, pt)
    g = rng.randn(*pt.shape).astype(F32)
    assert np.array_equal(cn.maxpool_backward(g, slot, x.shape), cl.maxpool_backward(g, arg, x.shape))


@pytest.mark.parametrize('n,expect', [(1, 1), (2, 1), (3, 2), (4, 2), (5, 3), (225, 113), (113, 57),
                                      (57, 29), (29, 15), (300, 150), (75, 38)])
def test_pooled_size_is_caffe_ceil_mode(n, expect):
    assert cn.pooled_size(n) == expect
    assert cl.lib().ref_pooled_size(n) == expect


def test_maxpool_first_max_wins_on_ties():
    x = np.zeros((1, 3, 3), F32)          # all equal: first element of each window in row-major order
    y, slot = cn.maxpool_forward(x)
    assert y.shape == (1, 2, 2) and np.all(slot == 0)
    dx = cn.maxpool_backward(np.array([[[1, 2], [3, 4]]], F32), slot, x.shape)
    assert np.array_equal(dx[0], np.array([[1, 0, 2], [0, 0, 0], [3, 0, 4]], F32))
    x = np.array([[[1, 5, 5], [5, 1, 0]]], F32)  # tie between (0,1) and (1,0): (0,1) scanned first
    y, slot = cn.maxpool_forward(x)
    dx = cn.maxpool_backward(np.ones_like(y), slot, x.shape)
    assert np.array_equal(dx[0], np.array([[0, 1, 1], [0, 0, 0]], F32))
    _, arg = cl.maxpool_forward(x)
    assert np.array_equal(cl.maxpool_backward(np.ones_like(y), arg, x.shape), dx)


def _torch_features(net, x):
    """Plain torch re-computation of the blobs, returning leaf tensors for autograd checks."""
    t = torch.tensor(x, dtype=torch.float32)
    blobs = {'data': t}
    for layer in net.topology:
        if layer[0] == 'conv':
            w, b = net.params[layer[1]]
            t = torch.relu(Fn.conv2d(t, torch.tensor(w), torch.tensor(b), padding=1))
        else:
            t = Fn.max_pool2d(t, 2, 2, ceil_mode=True)
        blobs[layer[1]] = t
    return blobs


def test_forward_blobs_are_post_relu_and_match_torch():
    topo = oracle.tiny_topology((8, 16), (2, 2), final_pool=True)
    net = oracle.NetOracle(topo, oracle.he_init_weights(topo, 3, bias_std=0.2))
    x = np.random.RandomState(0).randn(1, 3, 13, 10).astype(F32) * 40
    feats = net.forward(x)
    ref = _torch_features(net, x)
    assert list(feats) == net.layers() == ['data', 'conv1_1', 'conv1_2', 'pool1', 'conv2_1', 'conv2_2', 'pool2']
    for name, f in feats.items():
        assert f.shape == ref[name].shape
        assert np.allclose(f, ref[name].numpy(), rtol=1e-5, atol=1e-3), name
        if name.startswith('conv'):
            assert f.min() >= 0 and (f == 0).any()
        assert net.blob_shape(name, 13, 10) == f.shape[1:]


def test_backward_injection_is_unmasked_at_start_and_masked_from_above():
    """worker.py:88-106 + pycaffe ranges: diff_L = mask_L(incoming) + injected_L.

    Checked against torch autograd by expressing the rule as a surrogate loss:
    sum(inject_L * pre-ReLU-bypassed blob) -- i.e. the injected term must NOT see relu_L's mask,
    so we differentiate sum(inj * conv_out_L) with conv_out_L taken BEFORE the ReLU."""
    topo = oracle.tiny_topology((6, 8), (2, 2), final_pool=True)
    net = oracle.NetOracle(topo, oracle.he_init_weights(topo, 5, bias_std=0.3))
    rng = np.random.RandomState(1)
    x = (rng.randn(1, 3, 9, 12) * 30).astype(F32)
    feats = net.forward(x)
    names = ['pool2', 'conv2_1', 'conv1_2', 'pool1', 'data']
    diffs = {n: rng.randn(*feats[n].shape).astype(F32) for n in names}
    got = net.backward(diffs)

    xt = torch.tensor(x, requires_grad=True)
    t, surrogate = xt, (torch.tensor(diffs['data']) * xt).sum()
    for layer in net.topology:
        if layer[0] == 'conv':
            w, b = net.params[layer[1]]
            pre = Fn.conv2d(t, torch.tensor(w), torch.tensor(b), padding=1)
            if layer[1] in diffs:
                surrogate = surrogate + (torch.tensor(diffs[layer[1]]) * pre).sum()
            t = torch.relu(pre)
        else:
            t = Fn.max_pool2d(t, 2, 2, ceil_mode=True)
            if layer[1] in diffs:
                surrogate = surrogate + (torch.tensor(diffs[layer[1]]) * t).sum()
    surrogate.backward()
    assert np.allclose(got, xt.grad.numpy(), rtol=1e-4, atol=1e-4)

    # and it differs from the "plain autograd" answer that masks the injected diff too
    masked = {n: (d * (feats[n] > 0) if n.startswith('conv') else d) for n, d in diffs.items()}
    assert not np.allclose(net.backward(masked), got, rtol=1e-3, atol=1e-3)


def test_backward_hand_case_single_conv():
    """1 conv layer, identity-like weights: injected negative-side diff passes although blob == 0."""
    topo = (('conv', 'conv1_1', 3, 3),)
    w = np.zeros((3, 3, 3, 3), F32)
    for c in range(3):
        w[c, c, 1, 1] = 1.0
    net = oracle.NetOracle(topo, {'conv1_1': (w, np.zeros(3, F32))})
    x = -np.ones((1, 3, 2, 2), F32)          # conv output negative -> blob is 0 after ReLU
    f = net.forward(x)['conv1_1']
    assert np.all(f == 0)
    d = np.arange(12, dtype=F32).reshape(1, 3, 2, 2)
    assert np.array_equal(net.backward({'conv1_1': d}), d)     # unmasked
    assert np.array_equal(net.backward({}), np.zeros_like(x))


def test_full_forward_flag_changes_nothing():
    topo = oracle.tiny_topology((4, 4), (1, 1), final_pool=True)
    x = np.random.RandomState(2).randn(1, 3, 6, 6).astype(F32)
    a = oracle.NetOracle(topo, seed=1, full_forward=True).forward(x, ['conv1_1'])
    b = oracle.NetOracle(topo, seed=1, full_forward=False).forward(x, ['conv1_1'])
    assert np.array_equal(a['conv1_1'], b['conv1_1'])


def test_pre_and_deprocess_roundtrip():
    net = oracle.NetOracle(oracle.tiny_topology())
    img = np.random.RandomState(0).randint(0, 256, (5, 4, 3)).astype(np.uint8)
    x = net.preprocess(img)
    assert x.shape == (1, 3, 5, 4) and x.dtype == F32
    assert np.allclose(x[0, :, 0, 0], img[0, 0].astype(F32) - np.array([123.68, 116.779, 103.939], F32))
    assert np.allclose(net.deprocess(x), img, atol=1e-4)
