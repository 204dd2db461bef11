"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def tiny_setup(g):
    """(topology, params, weights, content, style, init) of tests/golden/transfer_tiny.npz."""
    import oracle
    tiny = json.loads(str(g['tiny_json']))
    topo = oracle.tiny_topology(**tiny)
    params = oracle.he_init_weights(topo, seed=0, bias_std=0.1)
    weights = json.loads(str(g['weights_json']))
    return topo, params, weights, g['content'], g['style'], g['init']


def check_trace(keys, vals, got, rtol, skip=('time',)):
    assert list(got.keys()) == [str(k) for k in keys]
    for k, v in zip(keys, vals):
        k = str(k)
        if k in skip:
            continue
        assert np.isclose(got[k], v, rtol=rtol, atol=1e-12), (k, got[k], v)


def trained_like_weights(topology, seed=0, gain_spread=100.0, dead_frac=0.05, bias_std=0.3):
    """Weights with the statistics of a TRAINED network rather than of an initialisation (the real vgg19.caffemodel cannot be
    fetched offline, reference download_models.sh:3-10): He-normal filters times a per-output-channel gain that is log-uniform
    over `gain_spread` (normalised to unit mean square, so activations keep their scale through the depth), biases N(0, bias_std),
    and `dead_frac` of the channels dead (zero filter, negative bias: the blob is zero there after the in-place ReLU)."""
    import oracle
    rng = np.random.RandomState(1000 + seed)
    params = oracle.he_init_weights(topology, seed=seed)
    out = type(params)()
    for name, (w, b) in params.items():
        cout = w.shape[0]
        gain = np.exp(rng.uniform(-0.5, 0.5, cout) * np.log(gain_spread))
        gain /= np.sqrt(np.mean(gain ** 2))
        w = (w * gain.reshape(-1, 1, 1, 1)).astype(np.float32)
        b = (rng.randn(cout) * bias_std).astype(np.float32)
        dead = rng.rand(cout) < dead_frac
        w[dead] = 0
        b[dead] = -np.abs(b[dead]) - 0.1
        out[name] = (w, b)
    return out


def receptive_geometry(topology):
    """Per blob (0 = data): (stride, lo, hi) such that unit y of the blob sees image rows [stride * y - lo, stride * y + hi]
    (3x3 / pad 1 convolutions, 2x2 / stride 2 pools; same along x)."""
    geo = [(1, 0, 0)]
    a, lo, hi = 1, 0, 0
    for layer in topology:
        if layer[0] == 'conv':
            lo, hi = lo + a, hi + a
        else:
            hi, a = hi + a, a * 2
        geo.append((a, lo, hi))
    return geo


def paint_receptive_fields(mask, positions, geom):
    """mask (H, W) bool: set the image-space receptive field of every blob position (y, x) in `positions` ((n, 2) ints)."""
    a, lo, hi = geom
    h, w = mask.shape
    for y, x in positions:
        mask[max(0, a * y - lo):min(h, a * y + hi + 1), max(0, a * x - lo):min(w, a * x + hi + 1)] = True
    return mask
