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
