#!/usr/bin/env python3
"""Generates tests/golden/*.npz|json by RUNNING THE REFERENCE ITSELF in the build container.

Run once, here (``python tests/golden/make_golden.py``); the fixtures are committed, this script
stays as their provenance.  /root/reference never travels to the GPU box -- only these vectors do.

What is imported unmodified from /root/reference: ``utils`` (tv_norm, p_norm, DecayingMean, dot,
axpy, Trace, resize_to_fit), ``optimizers`` (AdamOptimizer, LBFGSOptimizer), ``messages`` and
``worker`` (StyleTransfer, gram_matrix, Worker.process_message).  ``worker.py`` needs ``zmq`` at
import time (worker.py:13,24); pyzmq is not installed, so a stub module object is registered
first.  The network behind StyleTransfer is injected (worker.py:121-122 takes any model): it is
oracle.NetOracle, because Caffe and its weights are unavailable -- these vectors therefore pin
everything EXCEPT the conv/pool arithmetic (that is pinned in tests/test_oracle_net.py).
"""

import json
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

# --- import the reference -------------------------------------------------------------------
_zmq = types.ModuleType('zmq')
_zmq.Context = lambda: None
_zmq.PULL, _zmq.PUSH, _zmq.NOBLOCK = 7, 8, 1
_zmq.ZMQError = type('ZMQError', (Exception,), {})
sys.modules['zmq'] = _zmq
sys.path.insert(0, REF)
import messages as ref_messages          # noqa: E402
import optimizers as ref_optimizers      # noqa: E402
import utils as ref_utils                # noqa: E402
import worker as ref_worker              # noqa: E402
sys.path.remove(REF)

import oracle                            # noqa: E402

F32 = np.float32


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print('wrote', name, sorted(arrays))


# (1) tv_norm / p_norm ------------------------------------------------------------------------
def golden_image_norms():
    out = {}
    rng = np.random.RandomState(11)
    for tag, shape in (('a', (1, 3, 5, 7)), ('b', (1, 3, 16, 20))):
        x = (rng.randn(*shape) * 50).astype(F32)
        out['x_' + tag] = x
        for beta in (2, 1.5):
            v, g = ref_utils.tv_norm(x / 255, beta)
            out['tv_%s_%s_value' % (tag, beta)] = np.asarray(v)
            out['tv_%s_%s_grad' % (tag, beta)] = g
        for p in (2, 6):
            v, g = ref_utils.p_norm(x / 255, p)
            out['p_%s_%s_value' % (tag, p)] = np.asarray(v)
            out['p_%s_%s_grad' % (tag, p)] = g
    save('image_norms.npz', **out)


# (2) gram_matrix ------------------------------------------------------------------------------
def golden_gram():
    rng = np.random.RandomState(12)
    f = rng.randn(1, 8, 6, 5).astype(F32)
    save('gram.npz', feat=f, gram=ref_worker.gram_matrix(f))


# (3) DecayingMean, Adam, L-BFGS on a seeded quadratic -------------------------------------------
def quadratic(seed, shape):
    rng = np.random.RandomState(seed)
    n = int(np.prod(shape))
    a = rng.randn(n, n).astype(F32)
    a = (a @ a.T / n + np.eye(n, dtype=F32)).astype(F32)
    b = rng.randn(n).astype(F32)

    def opfunc(x):
        v = x.ravel()
        av = a @ v
        return F32(0.5) * np.dot(v, av) - np.dot(b, v), (av - b).reshape(x.shape)
    return a, b, opfunc


def golden_descent():
    out = {}
    rng = np.random.RandomState(13)
    items = rng.randn(6, 4).astype(F32)
    out['ema_items'] = items
    for decay in (0.9, 0.999):
        dm = ref_utils.DecayingMean(decay)
        seq = []
        for i, it in enumerate(items):
            seq.append(dm(it))
            if i == 3:
                dm.clear()
        out['ema_seq_%s' % decay] = np.stack(seq)

    shape = (1, 3, 4, 5)
    a, b, opfunc = quadratic(14, shape)
    out['quad_a'], out['quad_b'] = a, b
    x0 = np.random.RandomState(15).randn(*shape).astype(F32)
    out['x0'] = x0

    x = x0.copy()
    opt = ref_optimizers.AdamOptimizer(x, opfunc, step_size=0.1)
    xs, losses = [], []
    for i in range(7):
        if i == 4:
            opt.objective_changed()
        _, loss = opt.step()
        xs.append(x.copy())
        losses.append(loss)
    out['adam_xs'], out['adam_losses'] = np.stack(xs), np.asarray(losses, F32)

    x = x0.copy()
    opt = ref_optimizers.LBFGSOptimizer(x, opfunc, step_size=0.5)
    xs, losses, npairs = [], [], []
    for i in range(15):
        _, loss = opt.step()
        xs.append(x.copy())
        losses.append(loss)
        npairs.append(len(opt.sk))
    out['lbfgs_xs'], out['lbfgs_losses'] = np.stack(xs), np.asarray(losses, F32)
    out['lbfgs_npairs'] = np.asarray(npairs)
    save('descent.npz', **out)


# (4)+(5) StyleTransfer on a tiny injected model ---------------------------------------------------
TINY = dict(widths=(8, 16), convs_per_stage=(2, 2))
WEIGHTS = {'content': {'conv2_2': 0.08, 'conv1_2': 0.5},
           'style': {'conv1_1': 1, 'conv2_1': 1, 'conv1_2': 0.3, 'pool1': 0.7},
           'deepdream': {'conv2_1': 0.02}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
PARAMS_ODD = {'p': 3, 'p_power': 2, 'tv': 2, 'tv_power': 1.5}


def tiny_images():
    content = np.random.RandomState(1).randint(0, 256, (16, 20, 3)).astype(np.uint8)
    style = np.random.RandomState(2).randint(0, 256, (12, 12, 3)).astype(np.uint8)
    init = np.random.RandomState(3).randint(0, 256, (16, 20, 3)).astype(np.uint8)
    return content, style, init


def reference_transfer(optimizer, step, params):
    """Drives the reference's own message handler (worker.py:366-409) on a socket-less Worker."""
    topo = oracle.tiny_topology(**TINY)
    model = oracle.NetOracle(topo, oracle.he_init_weights(topo, seed=0, bias_std=0.1))
    wk = ref_worker.Worker.__new__(ref_worker.Worker)
    wk.transfer = ref_worker.StyleTransfer(model)
    wk.sock_out = types.SimpleNamespace(send_pyobj=lambda m: None)
    content, style, init = tiny_images()
    wk.process_message(ref_messages.SetImages(None, init, content, style, True))
    wk.process_message(ref_messages.SetWeights(WEIGHTS, params))
    wk.process_message(ref_messages.SetOptimizer(optimizer, step))
    wk.process_message(ref_messages.StartIteration())
    assert wk.transfer.is_running
    return wk.transfer


def trace_arrays(prefix, data, out):
    keys = [k for k in data]
    out[prefix + '_keys'] = np.asarray(keys)
    out[prefix + '_vals'] = np.asarray([0.0 if k == 'time' else float(data[k]) for k in keys])


def golden_transfer():
    content, style, init = tiny_images()
    out = dict(content=content, style=style, init=init,
               weights_json=np.asarray(json.dumps(WEIGHTS)),
               params_json=np.asarray(json.dumps(PARAMS)),
               params_odd_json=np.asarray(json.dumps(PARAMS_ODD)),
               tiny_json=np.asarray(json.dumps(TINY)))

    # (4) two opfunc evaluations: norm capture, then frozen norms
    for tag, params in (('std', PARAMS), ('odd', PARAMS_ODD)):
        st = reference_transfer('adam', 10, params)
        x = st.input.copy()
        for ev in (1, 2):
            out['%s_eval%d_x' % (tag, ev)] = x.copy()
            loss, grad = st.opfunc(x)
            out['%s_eval%d_loss' % (tag, ev)] = np.asarray(loss, F32)
            out['%s_eval%d_grad' % (tag, ev)] = grad
            trace_arrays('%s_eval%d_trace' % (tag, ev), st.traces[-1].data, out)
            x = x + F32(3.0) * np.sign(grad)
        for kind in 'csd':
            for layer, v in st.norms[kind].items():
                out['%s_norm_%s_%s' % (tag, kind, layer)] = np.asarray(v, F32)
        out[tag + '_layer_order'] = np.asarray(list(st.weights.index))
        out[tag + '_loss_nograd'] = np.asarray(st.opfunc(st.input, return_grad=False), F32)

    # (5) trajectories
    for name, kind, step, n in (('adam', 'adam', 10, 50), ('lbfgs', 'lbfgs', 1, 20)):
        st = reference_transfer(kind, step, PARAMS)
        losses, grads, images = [], [], []
        for i in range(n):
            image, trace = st.step()
            losses.append(trace['loss'])
            grads.append(trace['grad'])
            if i in (0, 4, n - 1):
                images.append(np.asarray(image, F32))
        out[name + '_losses'] = np.asarray(losses)
        out[name + '_grad_rms'] = np.asarray(grads)
        out[name + '_images'] = np.stack(images)
        trace_arrays(name + '_last_trace', trace, out)
    save('transfer_tiny.npz', **out)


# (6) DataFrame.from_dict layer order / NaN behaviour ------------------------------------------------
def golden_weight_order():
    import pandas as pd
    import yaml
    cases = {
        'initial_weights': yaml.safe_load(open(os.path.join(REF, 'initial_weights.yaml')))[0],
        'tiny': WEIGHTS,
        'interleaved': {'content': {'conv4_2': 1, 'conv2_2': 0},
                        'style': {'conv1_1': 1, 'conv4_2': 2, 'data': 1e-16},
                        'deepdream': {'pool3': -1}},
    }
    out = {}
    for name, weights in cases.items():
        df = pd.DataFrame.from_dict(weights, dtype=np.float32)
        nonzeros = abs(df) > 1e-15
        active = list(df.index[abs(nonzeros.sum(axis=1)) > 1e-15])
        out[name] = dict(weights=weights, rows=list(df.index), active=active,
                         cells={k: {r: (None if np.isnan(df[k][r]) else float(df[k][r]))
                                    for r in df.index} for k in df.columns})
    with open(os.path.join(HERE, 'weight_order.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote weight_order.json')


# (7) message pickles ----------------------------------------------------------------------------------
def golden_messages():
    m = ref_messages
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    from collections import OrderedDict
    objs = OrderedDict([
        ('SetImages', m.SetImages(None, img, img, m.SetImages.RESAMPLE, True)),
        ('SetImagesResample', m.SetImages((4, 6), m.SetImages.RESAMPLE, m.SetImages.RESAMPLE)),
        ('SetOptimizer', m.SetOptimizer('adam')),
        ('SetOptimizerStep', m.SetOptimizer('lbfgs', 0.5)),
        ('SetWeights', m.SetWeights(WEIGHTS, PARAMS)),
        ('StartIteration', m.StartIteration()),
        ('PauseIteration', m.PauseIteration()),
        ('Shutdown', m.Shutdown()),
        ('WorkerReady', m.WorkerReady(['data', 'conv1_1'])),
        ('Iterate', m.Iterate(img.astype(F32), 3, OrderedDict(loss=1.5, fevals=3))),
        ('GetImages', m.GetImages()),
    ])
    blob = {k: pickle.dumps(v, protocol=pickle.DEFAULT_PROTOCOL).hex() for k, v in objs.items()}
    with open(os.path.join(HERE, 'message_pickles.json'), 'w') as f:
        json.dump(blob, f, indent=1)
    print('wrote message_pickles.json')


# (8) config-1 inputs: resize_to_fit of the example images -----------------------------------------------
def golden_config1_inputs():
    from PIL import Image
    out, src = {}, {}
    for name in ('golden_gate', 'starry_night'):
        im = Image.open(os.path.join(REF, 'examples', name + '.jpg')).convert('RGB')
        src[name] = np.asarray(im)                  # the decoded pixels the app starts from (app.py:244-262): input of jobs.resize_to_fit
        out[name] = np.asarray(ref_utils.resize_to_fit(im, 256))
        print(name, im.size, '->', out[name].shape)
    save('config1_inputs.npz', **out)
    save('config1_sources.npz', **src)


if __name__ == '__main__':
    golden_image_norms()
    golden_gram()
    golden_descent()
    golden_transfer()
    golden_weight_order()
    golden_messages()
    golden_config1_inputs()
