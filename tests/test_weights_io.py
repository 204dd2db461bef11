"""Weight sources of the worker: .npz, the protobuf .caffemodel reader and the seeded
synthetic initialisation shared by tests and bench."""
import numpy as np
import pytest

import oracle
from style_transfer2_amd import caffemodel, weights
from style_transfer2_amd.engine import VGG19_TOPOLOGY

TOPO = (('conv', 'conv1_1', 3, 8), ('conv', 'conv1_2', 8, 8), ('pool', 'pool1'), ('conv', 'conv2_1', 8, 16))


@pytest.mark.parametrize('v1,legacy,packed', [(False, False, True), (True, True, True), (False, True, False),
                                              (True, False, False)])
def test_caffemodel_roundtrip_all_encodings(tmp_path, v1, legacy, packed):
    params = weights.he_normal(TOPO, seed=4, bias_std=0.3)
    path = tmp_path / 'net.caffemodel'
    caffemodel.write_caffemodel(str(path), params, v1=v1, legacy_dims=legacy, packed=packed)
    layers = caffemodel.read_caffemodel(str(path))
    assert list(layers) == list(params)
    got = caffemodel.vgg_params(layers, TOPO)
    for name, (w, b) in params.items():
        assert got[name][0].shape == w.shape and np.array_equal(got[name][0], w)
        assert np.array_equal(got[name][1], b)
    flipped = caffemodel.vgg_params(layers, TOPO, bgr_to_rgb=True)
    assert np.array_equal(flipped['conv1_1'][0], params['conv1_1'][0][:, ::-1])
    assert np.array_equal(flipped['conv1_2'][0], params['conv1_2'][0])


def test_caffemodel_errors():
    with pytest.raises(KeyError):
        caffemodel.vgg_params({}, TOPO)
    with pytest.raises(ValueError):
        caffemodel.read_caffemodel(b'\x0a\xff\xff\xff\xff\x0f')     # length runs past the end


def test_npz_roundtrip_and_seeded_init_matches_oracle(tmp_path):
    params = weights.he_normal(TOPO, seed=0, bias_std=0.1)
    path = str(tmp_path / 'w.npz')
    weights.save_npz(path, params)
    back = weights.load_npz(path, TOPO)
    for name in params:
        assert np.array_equal(back[name][0], params[name][0]) and np.array_equal(back[name][1], params[name][1])
    ref = oracle.he_init_weights(TOPO, seed=0, bias_std=0.1)      # same RandomState recipe on both sides
    for name in params:
        assert np.array_equal(ref[name][0], params[name][0]) and np.array_equal(ref[name][1], params[name][1])
    big = weights.he_normal(VGG19_TOPOLOGY, seed=0)
    assert list(big) == [l[1] for l in VGG19_TOPOLOGY if l[0] == 'conv'] and big['conv5_4'][0].shape == (512, 512, 3, 3)


def test_app_side_image_helpers_match_reference_fixture():
    """fit_into_square / resize_to_fit reproduce the reference's own resize of its example images
    (tests/golden/config1_inputs.npz was produced by reference utils.resize_to_fit)."""
    from style_transfer2_amd import jobs
    assert jobs.fit_into_square((979, 734), 256, True) == (256, 192)
    assert jobs.fit_into_square((1024, 640), 256, True) == (256, 160)
    assert jobs.fit_into_square((100, 50), 256) == (100, 50) and jobs.fit_into_square((100, 50), 256, True) == (256, 128)
    assert jobs.fit_into_square((50, 100), 30) == (15, 30)
    n = jobs.noise_image((4, 6), seed=1)
    assert n.shape == (4, 6, 3) and n.dtype == np.uint8 and np.array_equal(n, jobs.noise_image((4, 6), seed=1))
    from PIL import Image
    im = Image.fromarray(np.random.RandomState(0).randint(0, 256, (40, 60, 3)).astype(np.uint8))
    assert jobs.resize_to_fit(im, 30).size == (30, 20)


# ---------------------------------------------------------------- .caffemodel reader vs google.protobuf's own encoder
CAFFEMODEL_VARIANTS = ('v2_shape_packed', 'v1_legacy_unpacked', 'v2_double', 'v2_extra_fields')


@pytest.mark.parametrize('variant', CAFFEMODEL_VARIANTS)
def test_caffemodel_reader_on_protobuf_encoded_fixtures(variant, golden_dir):
    """tests/golden/caffemodel_*.bin were serialized by google.protobuf from a hand-written minimal caffe.proto schema
    (tests/golden/make_caffemodel_fixture.py): an encoder that shares nothing with caffemodel.py's test writer.
    V1 + V2 layers, BlobShape + legacy dims, packed + unpacked floats, double_data, unknown fields to skip."""
    import os
    raw = open(os.path.join(golden_dir, 'caffemodel_%s.bin' % variant), 'rb').read()
    params = weights.he_normal(TOPO, seed=4, bias_std=0.3)
    layers = caffemodel.read_caffemodel(raw)
    assert list(layers) == list(params), list(layers)          # weight-less layers (ReLU) are not reported
    got = caffemodel.vgg_params(layers, TOPO)
    for name, (w, b) in params.items():
        assert got[name][0].shape == w.shape and got[name][0].dtype == np.float32
        assert np.array_equal(got[name][0], w) and np.array_equal(got[name][1], b), (variant, name)


@pytest.mark.parametrize('variant', CAFFEMODEL_VARIANTS)
def test_caffemodel_fixtures_are_what_protobuf_encodes_today(variant, golden_dir):
    """The committed bytes are reproducible from the committed generator (skipped where google.protobuf is absent)."""
    import importlib.util
    import os
    pytest.importorskip('google.protobuf')
    spec = importlib.util.spec_from_file_location('make_caffemodel_fixture', os.path.join(golden_dir, 'make_caffemodel_fixture.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    raw = gen.encode(weights.he_normal(TOPO, seed=4, bias_std=0.3), variant)
    assert raw == open(os.path.join(golden_dir, 'caffemodel_%s.bin' % variant), 'rb').read()


# ---------------------------------------------------------------- network definition (config key `prototxt`)
def test_prototxt_roundtrip_and_refusals():
    """style_transfer2_amd/prototxt.py reads the subset of Caffe's text format that models/vgg19.prototxt uses (reference
    config.ini:28, worker.py:58-61) and refuses everything the engine would not run exactly."""
    from style_transfer2_amd import prototxt
    assert prototxt.parse(prototxt.write(VGG19_TOPOLOGY)) == VGG19_TOPOLOGY
    assert prototxt.parse(prototxt.write(TOPO)) == TOPO
    text = prototxt.write(TOPO)
    for bad, why in ((text.replace('kernel_size: 3', 'kernel_size: 5', 1), 'not a 3x3'),
                     (text.replace('pool: MAX', 'pool: AVE'), 'MAX pool'),
                     (text.replace('type: "ReLU"', 'type: "Sigmoid"', 1), 'in-place ReLU'),
                     (text + 'layer { bottom: "conv2_1" top: "fc" name: "fc" type: "InnerProduct" }', 'not supported'),
                     (text.replace('bottom: "conv1_1"\n    top: "conv1_2"', 'bottom: "data"\n    top: "conv1_2"'), 'linear chain'),
                     (text.replace('dim: 3', 'dim: 1'), 'N x 3 x H x W'),
                     (text + 'layer { name: "x" ', 'missing }')):
        with pytest.raises(ValueError, match=why):
            prototxt.parse(bad)
    # a convolution without its ReLU
    lines = text.split('layer {')
    no_relu = 'layer {'.join(l for l in lines if 'relu2_1' not in l)
    with pytest.raises(ValueError, match='in-place ReLU'):
        prototxt.parse(no_relu)


def test_prototxt_reader_on_the_reference_model_file():
    """The reference's own models/vgg19.prototxt (present in the build container only) parses to the built-in VGG19."""
    import os
    from style_transfer2_amd import prototxt
    path = '/root/reference/models/vgg19.prototxt'
    if not os.path.exists(path):
        pytest.skip('the reference tree is not on this machine')
    assert prototxt.read(path) == VGG19_TOPOLOGY
