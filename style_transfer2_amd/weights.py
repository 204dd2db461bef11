"""Where the conv weights come from.

The reference loads ``models/vgg19.caffemodel`` through pycaffe (worker.py:58-61, config.ini:28-29;
fetched by download_models.sh, unavailable offline).  Supported here:
  * ``.npz``  -- arrays ``<layer>_w`` (Cout,Cin,3,3) and ``<layer>_b`` (Cout,), RGB channel order;
  * seeded He-normal synthetic weights for tests, smoke and bench.
A ``.caffemodel`` protobuf reader is the first "next" row of SURVEY section 8f.
"""

from collections import OrderedDict

import numpy as np

F32 = np.float32


def he_normal(topology, seed=0, bias_std=0.0):
    """std = sqrt(2 / (9 Cin)), zero (or N(0, bias_std)) bias, numpy RandomState(seed) stream."""
    rng = np.random.RandomState(seed)
    params = OrderedDict()
    for layer in topology:
        if layer[0] != 'conv':
            continue
        _, name, cin, cout = layer
        w = (rng.randn(cout, cin, 3, 3) * np.sqrt(2.0 / (9 * cin))).astype(F32)
        b = (rng.randn(cout) * bias_std).astype(F32) if bias_std else np.zeros(cout, F32)
        params[name] = (w, b)
    return params


def load_npz(path, topology):
    data = np.load(path)
    params = OrderedDict()
    for layer in topology:
        if layer[0] == 'conv':
            params[layer[1]] = (np.asarray(data[layer[1] + '_w'], F32), np.asarray(data[layer[1] + '_b'], F32))
    return params


def save_npz(path, params):
    arrays = {}
    for name, (w, b) in params.items():
        arrays[name + '_w'], arrays[name + '_b'] = w, b
    np.savez(path, **arrays)
