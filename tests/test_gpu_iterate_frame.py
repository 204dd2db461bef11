"""Zero-copy iterates on the real engine (``pytest -m gpu``): the GPU copies each iterate into the payload slot of a
pre-formatted ``messages.Iterate`` pickle in pinned memory (st_step_frame_room, iterate_frame.py) and the worker sends that
buffer as it is.  What the app unpickles must be what the plain loop sends (reference worker.py:351-353, messages.py:64-74),
bit for bit, also across a re-allocation of the pinned buffers while the sender still holds earlier frames."""
from collections import deque
import pickle
import threading
import time

import numpy as np
import pytest

import messages
import oracle
import style_transfer2_amd as st2
import worker as worker_mod

pytestmark = pytest.mark.gpu
F32 = np.float32
WEIGHTS = {'content': {'conv2_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1}, 'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}


def _inputs():
    rs = np.random.RandomState
    return (rs(1).randint(0, 256, (32, 40, 3)).astype(np.uint8), rs(2).randint(0, 256, (24, 24, 3)).astype(np.uint8),
            rs(3).randint(0, 256, (32, 40, 3)).astype(np.uint8))


def _model():
    topo = oracle.tiny_topology((8, 16), (2, 2))
    return st2.HipModel(oracle.he_init_weights(topo, 0, 0.1), topology=topo)


class SlowFrameSockets:
    """In-process PULL/PUSH pair whose raw send behaves like libzmq's: the buffer is read when the tracker is waited on, a few
    milliseconds after send() returned (a slow wire: the worker runs ahead, the sender queue fills up)."""
    class Again(Exception):
        pass

    def __init__(self, script):
        self.inbound, self.sent, self.script, self.raw = deque(), [], dict(script), 0
        self.lock = threading.Lock()

    def recv_pyobj(self, flags=0):
        with self.lock:
            if not self.inbound:
                if flags:
                    raise self.Again()
                return messages.Shutdown()
            return pickle.loads(pickle.dumps(self.inbound.popleft()))

    def _note(self, obj):
        with self.lock:
            self.sent.append(obj)
            n_it = sum(isinstance(m, messages.Iterate) for m in self.sent)
            if isinstance(obj, messages.Iterate) and n_it in self.script:
                self.inbound.extend(self.script.pop(n_it))

    def send_pyobj(self, obj):
        self._note(pickle.loads(pickle.dumps(obj)))

    def send(self, data, copy=True, track=False):
        assert copy is False and track is True and isinstance(data, memoryview)
        self.raw += 1
        outer = self

        class Tracker:
            def wait(self):
                time.sleep(0.003)
                outer._note(pickle.loads(bytes(data)))
        return Tracker()


@pytest.mark.parametrize('async_iterate', ['1', '0'])
def test_zero_copy_worker_sends_the_plain_loops_iterates_across_an_upward_resample(async_iterate):
    content, style, init = _inputs()
    socks = SlowFrameSockets({4: [messages.SetImages((64, 80), messages.SetImages.RESAMPLE, messages.SetImages.RESAMPLE)],
                              9: [messages.PauseIteration()]})
    socks.inbound.extend([messages.SetImages(None, init, content, style, True), messages.SetWeights(WEIGHTS, PARAMS),
                          messages.SetOptimizer('adam', 10), messages.StartIteration()])
    wk = worker_mod.Worker({'async_iterate': async_iterate, 'zero_copy_iterate': '1'}, sock_in=socks, sock_out=socks,
                           transfer=st2.StyleTransfer(_model()))
    assert wk.pipelined and wk.zero_copy
    wk.run()
    kinds = [type(m).__name__ for m in socks.sent]
    its = [m for m in socks.sent if isinstance(m, messages.Iterate)]
    n = len(its)
    assert n >= 9 and kinds == ['WorkerReady'] + ['Iterate'] * n + ['Shutdown'] and socks.raw == n
    assert [m.i for m in its] == list(range(1, n + 1)) and [m.trace['fevals'] for m in its] == list(range(1, n + 1))
    shapes = [m.image.shape for m in its]
    k = shapes.index((64, 80, 3))                                   # iterates [0, k) came before the resample took effect
    assert 4 <= k <= 8 and shapes == [(32, 40, 3)] * k + [(64, 80, 3)] * (n - k)
    # the same job, synchronous, one owned array per step, the resample applied after the same iteration
    ref = st2.StyleTransfer(_model())
    ref.set_input(init); ref.set_content(content); ref.set_style(style); ref.reset()
    ref.set_weights(WEIGHTS, PARAMS)
    ref.optimizer_cls = st2.AdamOptimizer; ref.set_step_size(10); ref.reset()
    assert ref.start()
    for j, m in enumerate(its):
        if j == k:
            ref.resample_input((64, 80)); ref.resample_content((64, 80))
        img, tr = ref.step()
        assert m.image.dtype == F32 and np.array_equal(m.image, img), 'iterate %d differs' % m.i
        assert m.trace['loss'] == tr['loss'] and list(m.trace) == list(tr)


def test_views_handed_out_before_a_reallocation_keep_their_bytes():
    """st2.h: an iterate handed out by st_step_end stays valid for five further begins -- also when the pinned buffers are
    re-created in between because the input grew or the frame room changed (they are retired, not freed)."""
    content, style, init = _inputs()
    tr = st2.StyleTransfer(_model())
    tr.set_input(init); tr.set_content(content); tr.set_style(style); tr.reset()
    tr.set_weights(WEIGHTS, PARAMS)
    tr.optimizer_cls = st2.AdamOptimizer; tr.set_step_size(10); tr.reset()
    assert tr.start()
    held = []
    for _ in range(3):
        tr.step_begin()
        view, trace, index = tr.step_end(copy=False)
        held.append((view, view.copy()))
    # grow: RESAMPLE upward with the live optimizer, then frames on (other rooms): two re-allocations
    tr.resample_input((96, 128)); tr.resample_content((96, 128))
    tr.step_begin()
    big, _, _ = tr.step_end(copy=False)
    assert big.shape == (96, 128, 3)
    for view, copy in held:
        assert np.array_equal(view, copy)
    held.append((big, big.copy()))
    tr.enable_iterate_frames()
    for _ in range(4):                                              # four further begins: every view above is still inside its lifetime
        tr.step_begin()
        image, trace, index, frame = tr.step_end(frame=True)
        got = pickle.loads(frame)
        assert got.i == index and np.array_equal(got.image, image) and got.trace['loss'] == trace['loss']
    for view, copy in held[-1:]:
        assert np.array_equal(view, copy)


def test_worker_on_its_own_tcp_sockets_with_zero_copy_iterates(monkeypatch):
    """The deployment path end to end: Worker(config) creates its own PULL / PUSH sockets (reference worker.py:321-324), the engine
    copies every iterate into its pickle frame in pinned memory, the sender thread hands the frames to the transport with
    send(copy=False, track=True); an "app" across two TCP connections unpickles them with recv_pyobj.  pyzmq where it is installed,
    otherwise tests/minizmq.py (the same pickles framed per ZMTP 3.0).  The iterates are those of a synchronous run, bit for bit."""
    from test_boundary import zmq_module
    zmq, which = zmq_module(monkeypatch)
    print('[transport] GPU worker ran over', which)
    content, style, init = _inputs()
    ctx = zmq.Context()
    app_in = ctx.socket(zmq.PULL)
    port_app = app_in.bind_to_random_port('tcp://127.0.0.1')
    probe = ctx.socket(zmq.PULL)
    port_worker = probe.bind_to_random_port('tcp://127.0.0.1')
    probe.close(0)
    config = {'worker_socket': 'tcp://127.0.0.1:%d' % port_worker, 'app_socket': 'tcp://127.0.0.1:%d' % port_app}
    wk = worker_mod.Worker(config, transfer=st2.StyleTransfer(_model()))
    assert wk.pipelined and wk.zero_copy and isinstance(wk.sock_out, worker_mod.AsyncSender)
    th = threading.Thread(target=wk.run, daemon=True)
    th.start()
    app_out = ctx.socket(zmq.PUSH)
    app_out.connect(config['worker_socket'])
    for m in (messages.SetImages(None, init, content, style, True), messages.SetWeights(WEIGHTS, PARAMS),
              messages.SetOptimizer('adam', 10), messages.StartIteration()):
        app_out.send_pyobj(m)
    got = []
    poller = zmq.Poller()
    poller.register(app_in, zmq.POLLIN)
    while sum(isinstance(m, messages.Iterate) for m in got) < 8 and poller.poll(20000):
        got.append(app_in.recv_pyobj())
    app_out.send_pyobj(messages.Shutdown())
    while poller.poll(20000):
        got.append(app_in.recv_pyobj())
        if isinstance(got[-1], messages.Shutdown):
            break
    th.join(30)
    wk.close()
    app_out.close(0); app_in.close(0); ctx.destroy(0)
    kinds = [type(m).__name__ for m in got]
    its = [m for m in got if isinstance(m, messages.Iterate)]
    n = len(its)
    assert n >= 8 and kinds == ['WorkerReady'] + ['Iterate'] * n + ['Shutdown']
    assert [m.i for m in its] == list(range(1, n + 1))
    ref = st2.StyleTransfer(_model())
    ref.set_input(init); ref.set_content(content); ref.set_style(style); ref.reset()
    ref.set_weights(WEIGHTS, PARAMS)
    ref.optimizer_cls = st2.AdamOptimizer; ref.set_step_size(10); ref.reset()
    assert ref.start()
    for m in its:
        img, tr = ref.step()
        assert m.image.dtype == F32 and np.array_equal(m.image, img) and m.trace['loss'] == tr['loss'], m.i
