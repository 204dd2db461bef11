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


from oracle.receptive import receptive_geometry, paint_receptive_fields  # noqa: E402,F401  (shared with bench.py's parity leg)
