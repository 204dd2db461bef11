"""hipGraph replay of the steady-state Adam step (engine.cpp, step_graph_*): same kernels, same arguments, so the
trajectory must be BIT-identical to plain launches -- through reconfigurations that invalidate the captured graphs."""

import json
import os

import numpy as np
import pytest

import style_transfer2_amd as st2
from helpers import load, tiny_setup

pytestmark = pytest.mark.gpu


def make(graph, g):
    old = os.environ.get('ST2_GRAPH')
    os.environ['ST2_GRAPH'] = '1' if graph else '0'
    try:
        topo, net_params, weights, content, style, init = tiny_setup(g)
        st = st2.StyleTransfer(st2.HipModel(net_params, topology=topo))
    finally:
        if old is None:
            os.environ.pop('ST2_GRAPH', None)
        else:
            os.environ['ST2_GRAPH'] = old
    st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
    st.set_weights(weights, json.loads(str(g['params_json'])))
    st.optimizer_cls = st2.AdamOptimizer
    st.set_step_size(10)
    st.reset()
    assert st.start()
    return st, weights


def test_graph_replay_is_bit_identical_to_plain_launches():
    g = load('transfer_tiny.npz')
    a, weights = make(True, g)
    b, _ = make(False, g)
    other = np.random.RandomState(5).randint(0, 256, tuple(a.input_shape[2:]) + (3,)).astype(np.uint8)
    for i in range(40):
        if i == 12:                                   # device-resident stretch: nothing read back, graphs replayed
            for _ in range(7):
                a.step_async(); b.step_async()
        if i == 20:                                   # new weights: epoch change -> plain step, recapture
            w2 = {k: {n: v * 0.5 for n, v in d.items()} for k, d in weights.items()}
            for st in (a, b):
                st.set_weights(w2, json.loads(str(g['params_json'])))
        if i == 27:                                   # same-shape input replacement: m cleared (objective_changed)
            a.set_input(other); b.set_input(other)
        if i == 33:
            a.set_step_size(3); b.set_step_size(3)
        ia, ta = a.step()
        ib, tb = b.step()
        assert np.array_equal(ia, ib), i
        for k in ta:
            if k != 'time':
                assert ta[k] == tb[k], (i, k)
    assert b.engine.graph_replays() == 0
    assert a.engine.graph_replays() >= 25             # 47 steps, minus the plain step after each of the 4 reconfigurations


def test_graph_replay_matches_reference_adam_losses():
    g = load('transfer_tiny.npz')
    st, _ = make(True, g)
    losses = [st.step()[1]['loss'] for _ in range(50)]
    assert np.allclose(losses, g['adam_losses'], rtol=5e-3)
