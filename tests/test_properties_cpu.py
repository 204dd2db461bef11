"""Property tests (hypothesis) of two host-side data formats on the path's boundary:
  * the zero-copy ``Iterate`` frame must unpickle to what ``pickle.dumps(messages.Iterate(...))`` unpickles to for ANY image shape,
    iteration number and trace the worker can produce (reference worker.py:351-353, messages.py:64-74, utils.py:257-282);
  * ``prototxt.write`` / ``prototxt.parse`` round-trip every VGG-shaped linear topology, whatever the formatting noise."""
from collections import OrderedDict
import math
import pickle

from hypothesis import given, settings, strategies as st
import numpy as np

import messages
from style_transfer2_amd import iterate_frame, prototxt

F32 = np.float32

trace_values = st.one_of(
    st.floats(allow_nan=True, allow_infinity=True, width=64),
    st.integers(min_value=-2**62, max_value=2**62),
    st.floats(width=32).map(lambda v: float(np.float32(v))))
trace_keys = st.text(alphabet=st.characters(min_codepoint=33, max_codepoint=0x24f), min_size=1, max_size=40)


def same_scalar(a, b):
    if type(a) is not type(b):
        return False
    if isinstance(a, float) and math.isnan(a):
        return math.isnan(b)
    return a == b


@settings(max_examples=150, deadline=None)
@given(h=st.integers(1, 40), w=st.integers(1, 40), i=st.integers(0, 2**40),
       trace=st.lists(st.tuples(trace_keys, trace_values), max_size=60), seed=st.integers(0, 2**31 - 1))
def test_frame_equals_the_plain_pickle_for_any_iterate(h, w, i, trace, seed):
    image = np.random.RandomState(seed).randn(h, w, 3).astype(F32)
    tr = OrderedDict(trace)
    room = bytearray(iterate_frame.HEAD_ROOM + image.nbytes + iterate_frame.TAIL_ROOM)
    room[iterate_frame.HEAD_ROOM:iterate_frame.HEAD_ROOM + image.nbytes] = image.tobytes()
    frame = iterate_frame.assemble(room, iterate_frame.HEAD_ROOM, image.nbytes, image.shape, i, tr)
    got = pickle.loads(frame)
    ref = pickle.loads(pickle.dumps(messages.Iterate(image, i, tr), protocol=pickle.DEFAULT_PROTOCOL))
    assert type(got) is messages.Iterate and sorted(vars(got)) == ['i', 'image', 'trace']
    assert got.image.dtype == F32 and got.image.shape == ref.image.shape and np.array_equal(got.image, ref.image, equal_nan=True)
    assert type(got.i) is int and got.i == ref.i
    assert type(got.trace) is type(ref.trace) and list(got.trace) == list(ref.trace)
    assert all(same_scalar(a, b) for a, b in zip(got.trace.values(), ref.trace.values()))


@st.composite
def topologies(draw):
    topo, cin = [], 3
    stages = draw(st.integers(1, 5))
    for s in range(1, stages + 1):
        for j in range(1, draw(st.integers(1, 4)) + 1):
            cout = draw(st.sampled_from([3, 8, 16, 24, 64, 128, 512]))
            topo.append(('conv', 'conv%d_%d' % (s, j), cin, cout))
            cin = cout
        if s < stages or draw(st.booleans()):
            topo.append(('pool', 'pool%d' % s))
    return tuple(topo)


@settings(max_examples=100, deadline=None)
@given(topo=topologies(), noise=st.integers(0, 3))
def test_prototxt_round_trips_any_vgg_shaped_chain(topo, noise):
    text = prototxt.write(topo)
    if noise & 1:               # comments and blank lines anywhere between lines
        text = '\n'.join(line + ('   # a comment' if k % 3 == 0 else '') + ('\n' if k % 5 == 0 else '') for k, line in enumerate(text.split('\n')))
    if noise & 2:               # one statement per token run: the text format does not care about line breaks
        text = text.replace('\n', ' \n ').replace('{', ' {\n').replace('}', '\n}')
    assert prototxt.parse(text) == topo
