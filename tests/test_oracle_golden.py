"""The CPU oracle against vectors produced by the reference itself (tests/golden/make_golden.py).

These pin every part of the oracle except the conv/pool arithmetic (tests/test_oracle_net.py)."""
import json

import numpy as np
import pytest

import oracle
from oracle import descent, objective
from helpers import load, load_json, tiny_setup, check_trace, rel_l2

F32 = np.float32


def test_tv_and_p_terms_match_reference():
    g = load('image_norms.npz')
    for tag in 'ab':
        x = g['x_' + tag]
        for beta in (2, 1.5):
            v, grad = oracle.tv_term(x / 255, beta)
            assert np.array_equal(grad, g['tv_%s_%s_grad' % (tag, beta)])
            assert v == g['tv_%s_%s_value' % (tag, beta)]
        for p in (2, 6):
            v, grad = oracle.p_term(x / 255, p)
            assert np.array_equal(grad, g['p_%s_%s_grad' % (tag, p)])
            assert v == g['p_%s_%s_value' % (tag, p)]


def test_gram_matches_reference():
    g = load('gram.npz')
    assert np.array_equal(oracle.gram(g['feat']), g['gram'])


def test_ema_matches_reference():
    g = load('descent.npz')
    for decay in (0.9, 0.999):
        ema = descent.EmaBiasCorrected(decay)
        for i, item in enumerate(g['ema_items']):
            ema.push(item)
            assert np.array_equal(ema.value(), g['ema_seq_%s' % decay][i])
            if i == 3:
                ema.clear()


def _quadratic(g):
    a, b = g['quad_a'], g['quad_b']

    def opfunc(x):
        v = x.ravel()
        av = a @ v
        return F32(0.5) * np.dot(v, av) - np.dot(b, v), (av - b).reshape(x.shape)
    return opfunc


def test_adam_matches_reference_incl_objective_changed():
    g = load('descent.npz')
    x = g['x0'].copy()
    opt = descent.AdamOracle(x, _quadratic(g), step_size=0.1)
    for i in range(7):
        if i == 4:
            opt.objective_changed()
        _, loss = opt.step()
        assert np.array_equal(x, g['adam_xs'][i]), i
        assert loss == g['adam_losses'][i]


def test_lbfgs_matches_reference_incl_eviction():
    g = load('descent.npz')
    x = g['x0'].copy()
    opt = descent.LBFGSOracle(x, _quadratic(g), step_size=0.5)
    for i in range(15):
        _, loss = opt.step()
        assert np.array_equal(x, g['lbfgs_xs'][i]), i
        assert len(opt.pairs) == g['lbfgs_npairs'][i]
    assert g['lbfgs_npairs'].max() == 10     # n_corr eviction was exercised


def test_weight_table_matches_pandas():
    for name, case in load_json('weight_order.json').items():
        rows, cells = objective.weight_table(case['weights'])
        assert rows == case['rows'], name
        st = oracle.TransferOracle(oracle.NetOracle(oracle.tiny_topology()))
        st.rows, st.cells = rows, cells
        assert st.active_layers() == case['active'], name
        for kind, col in case['cells'].items():
            for layer, v in col.items():
                got = cells[kind][layer]
                assert (np.isnan(got) if v is None else float(got) == v), (name, kind, layer)


def _oracle_transfer(g, kind, step, params):
    topo, net_params, weights, content, style, init = tiny_setup(g)
    st = oracle.TransferOracle(oracle.NetOracle(topo, net_params))
    st.set_input(init)
    st.set_content(content)
    st.set_style(style)
    st.reset()
    st.set_weights(weights, params)
    st.set_optimizer(kind, step)
    assert st.start()
    return st


@pytest.mark.parametrize('tag', ['std', 'odd'])
def test_opfunc_two_evals_match_reference(tag):
    g = load('transfer_tiny.npz')
    params = json.loads(str(g['params_json' if tag == 'std' else 'params_odd_json']))
    st = _oracle_transfer(g, 'adam', 10, params)
    assert st.active_layers() == [str(s) for s in g[tag + '_layer_order']]
    for ev in (1, 2):
        loss, grad = st.opfunc(g['%s_eval%d_x' % (tag, ev)].copy())
        assert np.array_equal(grad, g['%s_eval%d_grad' % (tag, ev)])
        assert loss == g['%s_eval%d_loss' % (tag, ev)]
        check_trace(g['%s_eval%d_trace_keys' % (tag, ev)], g['%s_eval%d_trace_vals' % (tag, ev)],
                    st.traces[-1].data, rtol=0)
    for kind in 'csd':
        for layer, v in st.norms[kind].items():
            assert v == g['%s_norm_%s_%s' % (tag, kind, layer)]
    assert st.opfunc(st.input, return_grad=False) == g[tag + '_loss_nograd']


@pytest.mark.parametrize('name,kind,step,n', [('adam', 'adam', 10, 50), ('lbfgs', 'lbfgs', 1, 20)])
def test_trajectories_match_reference(name, kind, step, n):
    g = load('transfer_tiny.npz')
    st = _oracle_transfer(g, kind, step, json.loads(str(g['params_json'])))
    images = []
    for i in range(n):
        image, trace = st.step()
        assert trace['loss'] == g[name + '_losses'][i], i
        assert trace['grad'] == g[name + '_grad_rms'][i], i
        if i in (0, 4, n - 1):
            images.append(image)
    assert np.array_equal(np.stack(images), g[name + '_images'])
    check_trace(g[name + '_last_trace_keys'], g[name + '_last_trace_vals'], trace, rtol=0)
    assert rel_l2(images[-1], g[name + '_images'][-1]) == 0.0


def test_stored_oracle_trajectories_are_what_the_oracle_computes():
    """tests/golden/oracle_trajectories.npz holds whole CPU-oracle trajectories that the GPU tests compare the engine with
    (made by tests/golden/make_trajectories.py).  Their first steps, re-run here with the same functions: a change of the oracle
    that moved them would leave the GPU tests comparing against a stale answer."""
    import importlib.util
    import os
    from helpers import GOLDEN
    spec = importlib.util.spec_from_file_location('make_trajectories', os.path.join(GOLDEN, 'make_trajectories.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    g = load('oracle_trajectories.npz')
    l32, _ = mk.drift_run('fp32', 2)
    l16, _ = mk.drift_run('bf16', 1)
    lc, _, keys, vals = mk.config1_run(3)
    assert np.allclose(l32, g['drift_losses_fp32'][:2], rtol=1e-6, atol=0), (l32, g['drift_losses_fp32'][:2])
    assert np.allclose(l16, g['drift_losses_bf16'][:1], rtol=1e-6, atol=0), (l16, g['drift_losses_bf16'][:1])
    assert np.allclose(lc, g['config1_losses'][:3], rtol=1e-6, atol=0), (lc, g['config1_losses'][:3])
    assert keys == [str(k) for k in g['config1_first_keys']]
    stored = dict(zip(keys, g['config1_first_values']))
    for k, v in zip(keys, vals):
        if k != 'time':
            assert np.isclose(v, stored[k], rtol=1e-6), (k, v, stored[k])
    assert len(g['drift_losses_fp32']) == 11 and len(g['config1_losses']) == 50 and g['drift_final_bf16'].shape == (192, 256, 3)
    # the 768 x 1024 part (round 5): its first L-BFGS step (two objective evaluations at size, ~25 s here)
    ls, _ = mk.drift_run('fp32', 1, fit=1024)
    assert np.allclose(ls, g['size_losses_fp32'][:1], rtol=1e-6, atol=0), (ls, g['size_losses_fp32'][:1])
    assert len(g['size_losses_fp32']) == 5 and len(g['size_losses_bf16']) == 3 and g['size_final_fp32_sub4'].shape == (192, 256, 3)
