"""The device-resident L-BFGS (lbfgs.hip: one fused launch per link of the two-loop recursion, ring bookkeeping and
the s.y > 1e-10 gate on the device) against the CPU oracle's restatement of optimizers.py:49-125, which is itself pinned
bit-exactly to the reference's LBFGSOptimizer by tests/golden/descent.npz."""
import json

import numpy as np
import pytest

import oracle
import style_transfer2_amd as st2
from helpers import load, rel_l2, tiny_setup

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(params=['chain', 'gram'], autouse=True)
def lbfgs_form(request, monkeypatch):
    """Every test below runs on both forms of the device recursion (lbfgs.hip): the chain of fused axpy + dot links in the
    reference's operation order (the fp32 default) and the Gram form -- coefficients against the matrix of inner products, two
    passes over the history per step (the bf16 feature path's default).  Same mathematics, different rounding."""
    monkeypatch.setenv('ST2_LBFGS_FORM', request.param)
    return request.param


def _history(rng, shape, n_pairs):
    """(s, y, s.y) triples with positive curvature: y = D s + noise for a positive diagonal D."""
    d = (0.5 + rng.rand(*shape)).astype(F32)
    pairs = []
    for _ in range(n_pairs):
        s = rng.randn(*shape).astype(F32)
        y = (d * s + 0.05 * rng.randn(*shape)).astype(F32)
        pairs.append((s, y, oracle.descent.sdot(s, y)))
    return pairs


@pytest.mark.parametrize('h,w', [(15, 17), (64, 96), (225, 300)])      # 3hw % 4 = 1, 0, 0: the scalar tail and the float4 body
@pytest.mark.parametrize('n_pairs', [0, 1, 2, 5, 10])
def test_device_two_loop_matches_oracle_inv_hessian_times(h, w, n_pairs, lbfgs_form):
    rng = np.random.RandomState(100 * n_pairs + h)
    shape = (1, 3, h, w)
    pairs = _history(rng, shape, n_pairs)
    g = (rng.randn(*shape) * 3).astype(F32)
    ora = oracle.LBFGSOracle(np.zeros(shape, F32), None)
    ora.pairs = list(pairs)
    want = ora.inv_hessian_times(g)
    eng = st2.Engine(oracle.tiny_topology((8,), (1,)))
    eng.set_input(np.zeros((h, w, 3), np.uint8))
    got = eng.lbfgs_inv_hv([(s, y) for s, y, _ in pairs], g)
    assert got.shape == want.shape
    print('[inv_hv %s] %dx%d, %d pairs: rel-L2 %.2e' % (lbfgs_form, h, w, n_pairs, rel_l2(got, want)))
    assert rel_l2(got, want) <= 1e-5, rel_l2(got, want)
    again = eng.lbfgs_inv_hv([(s, y) for s, y, _ in pairs], g)
    assert np.array_equal(again, got)                                  # fixed-order reductions: bitwise reproducible
    eng.close()


def test_device_gate_rejects_pairs_without_positive_curvature():
    rng = np.random.RandomState(3)
    shape = (1, 3, 16, 20)
    s = rng.randn(*shape).astype(F32)
    eng = st2.Engine(oracle.tiny_topology((8,), (1,)))
    eng.set_input(np.zeros((16, 20, 3), np.uint8))
    with pytest.raises(st2.StError, match='gate'):
        eng.lbfgs_inv_hv([(s, -s)], s)                                 # s.y < 0: optimizers.py:82 drops the pair
    with pytest.raises(st2.StError, match='gate'):
        eng.lbfgs_inv_hv([(s, s), (s * F32(1e-8), s * F32(1e-8))], s)   # 0 < s.y <= 1e-10: dropped too
    eng.close()


def test_lbfgs_trajectory_per_step_against_reference_vectors():
    """20 fixed-step L-BFGS iterations on the golden tiny job (the reference's own LBFGSOptimizer produced the
    vectors; they pass through the n_corr = 10 eviction).  Per-step loss within 1e-3 for the first 10 steps."""
    g = load('transfer_tiny.npz')
    topo, net_params, weights, content, style, init = tiny_setup(g)
    st = st2.StyleTransfer(st2.HipModel(net_params, topology=topo))
    st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
    st.set_weights(weights, json.loads(str(g['params_json'])))
    st.optimizer_cls = st2.LBFGSOptimizer; st.set_step_size(1); st.reset()
    assert st.start()
    losses = []
    for i in range(20):
        image, trace = st.step()
        losses.append(trace['loss'])
    assert np.allclose(losses[:10], g['lbfgs_losses'][:10], rtol=1e-3), np.abs(np.array(losses[:10]) / g['lbfgs_losses'][:10] - 1).max()
    assert np.allclose(losses, g['lbfgs_losses'], rtol=5e-2)


def test_lbfgs_against_oracle_run_beside_it_with_objective_changed():
    """Oracle and device side by side: 14 steps (past the eviction), a set_input of equal shape in between
    (objective_changed wipes the history and the cached gradient, optimizers.py:121-125)."""
    g = load('transfer_tiny.npz')
    topo, net_params, weights, content, style, init = tiny_setup(g)
    params = json.loads(str(g['params_json']))
    ora = oracle.TransferOracle(oracle.NetOracle(topo, net_params))
    dev = st2.StyleTransfer(st2.HipModel(net_params, topology=topo))
    for st in (ora, dev):
        st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
        st.set_weights(weights, params)
    ora.set_optimizer('lbfgs', 1)
    dev.optimizer_cls = st2.LBFGSOptimizer; dev.set_step_size(1); dev.reset()
    assert ora.start() and dev.start()
    other = np.random.RandomState(7).randint(0, 256, init.shape).astype(np.uint8)
    for i in range(14):
        if i == 12:
            ora.set_input(other); dev.set_input(other)
        io, to = ora.step()
        idv, td = dev.step()
        tol = 1e-3 if i < 8 or i >= 12 else 2e-2
        assert np.isclose(td['loss'], to['loss'], rtol=tol), (i, td['loss'], to['loss'])
        assert list(td) == list(to)
    assert np.mean((idv - io) ** 2) <= 1e-2
