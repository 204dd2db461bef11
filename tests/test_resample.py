"""Pillow-exact resampling: the restated coefficient tables and two-pass float resize against Pillow itself
(the third-party library the reference calls, utils.py:131) on the CPU, and the device kernels on the GPU."""
import numpy as np
import pytest
from PIL import Image

from style_transfer2_amd import resample

F32 = np.float32


def same(a, b):
    """Bit-exact up to a rare final-rounding ulp: Pillow's C loop and the restatement agree exactly in this
    container; on some hosts one element in ~10^4 differs by 1 ulp of float32 (double->float rounding of a sum
    formed with / without a fused multiply-add in Pillow's build)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    d = np.abs(a - b)
    return bool(np.all(d <= 2.4e-7 * np.maximum(np.abs(b), 1e-30) + 1e-30) and np.mean(d > 0) <= 1e-3)


@pytest.mark.parametrize('hw_in,hw_out', [((13, 17), (7, 9)), ((13, 17), (20, 30)), ((16, 20), (16, 31)), ((16, 20), (9, 20)),
                                          ((32, 40), (48, 64)), ((225, 300), (150, 200)), ((5, 7), (5, 7))])
@pytest.mark.parametrize('method', [resample.LANCZOS, resample.BILINEAR])
def test_restated_resize_is_bit_exact_with_pillow(hw_in, hw_out, method):
    a = (np.random.RandomState(hw_in[0] * hw_out[1]).randn(2, 3, *hw_in) * 50).astype(F32)
    ref = resample.resample_nchw(a, hw_out, method)                 # Pillow, as the reference uses it
    got = resample.resample_planes_reference(a, hw_out, method)
    assert same(got, ref)
    lo, n, k = resample.pillow_coeffs(hw_in[1], hw_out[1], method)
    assert np.all(lo >= 0) and np.all(lo + n <= hw_in[1]) and np.allclose(k.sum(1), 1.0)


@pytest.mark.gpu
def test_device_resample_of_adam_state_matches_pillow_bit_for_bit():
    import oracle
    import style_transfer2_amd as st2
    topo = oracle.tiny_topology((8, 16), (2, 2))
    st = st2.StyleTransfer(st2.HipModel(oracle.he_init_weights(topo, 0, 0.1), topology=topo))
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (32, 40, 3)).astype(np.uint8), rs(2).randint(0, 256, (24, 24, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (32, 40, 3)).astype(np.uint8))
    st.set_input(init); st.set_content(content); st.set_style(style); st.reset()
    st.set_weights({'content': {'conv2_2': 0.08}, 'style': {'conv1_1': 1}, 'deepdream': {}}, {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2})
    st.optimizer_cls = st2.AdamOptimizer; st.set_step_size(10); st.reset()
    assert st.start()
    for _ in range(3):
        st.step()
    eng = st.engine
    for size in ((48, 64), (20, 26), (20, 40)):
        x = eng.get_input_nchw()
        m, v, i1, i2 = eng.adam_get_state()
        content_before = st.content
        want_x = resample.resample_nchw(x, size)
        want_m = resample.resample_nchw(m, size)
        want_v = np.maximum(0, resample.resample_nchw(v, size, method=resample.BILINEAR))
        want_c = resample.resample_nchw(content_before, size)
        st.resample_input(size)
        st.resample_content(size)
        m2, v2, j1, j2 = eng.adam_get_state()
        assert same(eng.get_input_nchw(), want_x)
        assert np.array_equal(m2, np.zeros_like(want_m))      # resample_input ends with objective_changed(): m cleared
        assert same(v2, want_v) and (j1, j2) == (0, i2)
        assert same(st.content, want_c)
        assert st.check_consistency()
        st.step()
    # the momentum itself, without the objective_changed() that resample_input adds
    m, v, i1, i2 = eng.adam_get_state()
    want_m = resample.resample_nchw(m, (30, 36))
    st.optimizer.resample((30, 36))
    m2, v2, j1, j2 = eng.adam_get_state()
    assert same(m2, want_m) and (j1, j2) == (i1, i2)
    # new_x path (set_input with a different shape and a live optimizer, worker.py:196-198)
    new = rs(9).randint(0, 256, (24, 28, 3)).astype(np.uint8)
    st.set_input(new)
    assert st.input_shape == (1, 3, 24, 28) and np.array_equal(eng.get_input_nchw(), st.model.preprocess(new))
