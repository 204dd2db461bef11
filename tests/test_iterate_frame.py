"""The zero-copy ``Iterate`` frame (style_transfer2_amd/iterate_frame.py): a pickle assembled around an image that is already
in (pinned) memory must unpickle to exactly what the reference's ``send_pyobj(Iterate(image, i, trace))`` would have produced
(reference worker.py:351-353, messages.py:64-74), and the worker's zero-copy path must keep the wire order."""
from collections import OrderedDict
import pickle
import pickletools
import threading
import time

import numpy as np
import pytest

import messages
import worker as worker_mod
from style_transfer2_amd import iterate_frame
from helpers import load_json
from test_boundary import FakePipelinedTransfer, FakeSockets, run_worker_over_tcp, zmq_module

F32 = np.float32


def framed(image, i, trace, head_room=iterate_frame.HEAD_ROOM, tail_room=iterate_frame.TAIL_ROOM):
    room = bytearray(head_room + image.nbytes + tail_room)
    room[head_room:head_room + image.nbytes] = image.tobytes()
    return iterate_frame.assemble(room, head_room, image.nbytes, image.shape, i, trace), room


def trace_of(n_layers=6):
    t = OrderedDict()
    for k in range(n_layers):
        for tag in ('c', 's', 'd'):
            t['conv%d_1_%s_loss' % (k, tag)] = 0.25 * k
            t['conv%d_1_%s_grad' % (k, tag)] = 1e-3 * k
    for k in ('scd_loss', 't_loss', 'p_loss', 'scd_grad', 't_grad', 'p_grad', 'time', 'loss', 'grad'):
        t[k] = float(len(t))
    t['fevals'] = 17
    return t


@pytest.mark.parametrize('shape', [(1, 1, 3), (5, 7, 3), (192, 256, 3)])
def test_frame_unpickles_to_what_send_pyobj_would_have_sent(shape):
    image = np.random.RandomState(3).randn(*shape).astype(F32)
    trace = trace_of()
    frame, _ = framed(image, 17, trace)
    got = pickle.loads(frame)
    ref = pickle.loads(pickle.dumps(messages.Iterate(image, 17, trace), protocol=pickle.DEFAULT_PROTOCOL))
    assert type(got) is messages.Iterate and sorted(vars(got)) == sorted(vars(ref)) == ['i', 'image', 'trace']
    assert got.image.dtype == ref.image.dtype == F32 and got.image.shape == ref.image.shape
    assert got.image.flags['C_CONTIGUOUS'] and got.image.flags['WRITEABLE'] == ref.image.flags['WRITEABLE']
    assert np.array_equal(got.image, image)
    assert type(got.i) is int and got.i == ref.i
    assert type(got.trace) is OrderedDict and list(got.trace.items()) == list(ref.trace.items())
    assert all(type(a) is type(b) for a, b in zip(got.trace.values(), ref.trace.values()))
    # the image bytes are IN the frame (one operand of the pickle program), not copied around it
    assert len(frame) - image.nbytes < 4096
    assert sum(1 for _ in pickletools.genops(bytes(frame))) > 20                                     # a well-formed stream


def test_frame_names_the_class_the_reference_pickles_name():
    """tests/golden/message_pickles.json holds the reference's own pickles: the class is referenced as messages.Iterate."""
    blob = bytes.fromhex(load_json('message_pickles.json')['Iterate'])
    frame, _ = framed(np.zeros((2, 2, 3), F32), 1, OrderedDict(loss=1.0))

    def globals_of(data):
        strings = [arg for op, arg, _ in pickletools.genops(data) if op.name in ('SHORT_BINUNICODE', 'BINUNICODE', 'GLOBAL')]
        return ' '.join(str(v) for v in strings)
    for data in (bytes(frame), blob):
        assert 'messages' in globals_of(data) and 'Iterate' in globals_of(data)
    a, b = pickle.loads(bytes(frame)), pickle.loads(blob)
    assert type(a) is type(b) and sorted(vars(a)) == sorted(vars(b))


def test_frame_room_is_checked_and_large_images_use_an_8_byte_length():
    image = np.zeros((2, 2, 3), F32)
    with pytest.raises(ValueError):
        framed(image, 1, trace_of(), tail_room=16)
    with pytest.raises(ValueError):
        framed(image, 1, trace_of(), head_room=64)
    big = iterate_frame.head((40000, 40000, 3))                      # 19.2 GB: BINBYTES8, protocol 4
    assert big[:2] == b'\x80\x04' and big[-9:-8] == b'\x8e' and int.from_bytes(big[-8:], 'little') == 40000 * 40000 * 12
    small = iterate_frame.head((1024, 1024, 3))
    assert small[:2] == b'\x80\x03' and small[-5:-4] == b'B' and int.from_bytes(small[-4:], 'little') == 1024 * 1024 * 12


# ------------------------------------------------------------------------------------------ the worker's zero-copy path
class FrameSockets(FakeSockets):
    """An outbound socket with pyzmq's raw ``send(data, copy=False, track=True)`` beside send_pyobj.  Like libzmq, it reads the
    buffer some time AFTER send returns unless the caller waits on the tracker: the bytes are captured in ``wait``."""
    def __init__(self, inbound):
        super().__init__(inbound)
        self.raw_sends = 0

    def send(self, data, copy=True, track=False):
        assert copy is False and track is True, 'the zero-copy path sends with copy=False, track=True'
        assert isinstance(data, memoryview)
        self.raw_sends += 1
        outer = self

        class Tracker:
            def wait(self_inner):
                outer.sent.append(pickle.loads(bytes(data)))
        return Tracker()


class FakeFramedTransfer(FakePipelinedTransfer):
    """A pipelined backend that also offers the finished pickle of each iterate in a rotating buffer (as the engine does)."""
    def __init__(self, max_steps=5):
        super().__init__(max_steps)
        self.frames_on = False
        self.rooms = [bytearray(iterate_frame.HEAD_ROOM + 48 + iterate_frame.TAIL_ROOM) for _ in range(6)]

    def enable_iterate_frames(self):
        self.frames_on = True

    def step_end(self, copy=True, frame=False):
        image, trace, i = super().step_end()
        if not frame:
            return image, trace, i
        assert self.frames_on
        room = self.rooms[i % 6]
        room[iterate_frame.HEAD_ROOM:iterate_frame.HEAD_ROOM + 48] = image.tobytes()
        return image, trace, i, iterate_frame.assemble(room, iterate_frame.HEAD_ROOM, 48, image.shape, i, trace)


@pytest.mark.parametrize('async_iterate', ['0', '1'])
def test_worker_zero_copy_iterates_keep_the_wire(async_iterate):
    img = np.zeros((4, 4, 3), np.uint8)
    socks = FrameSockets([messages.SetImages(None, img, img, img, True), messages.StartIteration()])
    tr = FakeFramedTransfer(5)
    wk = worker_mod.Worker({'async_iterate': async_iterate, 'zero_copy_iterate': '1'}, sock_in=socks, sock_out=socks, transfer=tr)
    assert wk.pipelined and wk.zero_copy and tr.frames_on
    wk.run()
    kinds = [type(m).__name__ for m in socks.sent]
    assert kinds == ['WorkerReady'] + ['Iterate'] * 5 + ['Shutdown']
    its = socks.sent[1:6]
    assert socks.raw_sends == 5
    assert [m.i for m in its] == [1, 2, 3, 4, 5] and [m.trace['fevals'] for m in its] == [1, 2, 3, 4, 5]
    assert [float(m.image[0, 0, 0]) for m in its] == [1, 2, 3, 4, 5] and its[0].image.shape == (2, 2, 3)


def test_injected_sockets_get_owned_copies_by_default():
    """Zero copy is for the pyzmq sockets the worker created itself (or an explicit config key): an injected socket that keeps
    the message objects must never be handed views of buffers the engine rewrites."""
    img = np.zeros((4, 4, 3), np.uint8)
    socks = FrameSockets([messages.SetImages(None, img, img, img, True), messages.StartIteration()])
    tr = FakeFramedTransfer(3)
    wk = worker_mod.Worker({}, sock_in=socks, sock_out=socks, transfer=tr)
    assert wk.pipelined and not wk.zero_copy and not tr.frames_on
    wk.run()
    assert socks.raw_sends == 0
    assert [type(m).__name__ for m in socks.sent] == ['WorkerReady'] + ['Iterate'] * 3 + ['Shutdown']


def test_zero_copy_frames_over_the_workers_own_tcp_sockets(monkeypatch):
    """The worker's default on sockets it created itself: every Iterate leaves as `send(frame, copy=False, track=True)` -- here over
    real TCP connections (pyzmq where installed, tests/minizmq.py's ZMTP 3.0 framing otherwise) and is unpickled by the app side's
    plain recv_pyobj."""
    zmq, which = zmq_module(monkeypatch)
    tr = FakeFramedTransfer(6)
    wk, got = run_worker_over_tcp(zmq, tr, 6)
    print('[transport] zero-copy iterates ran over', which)
    assert wk.zero_copy and tr.frames_on                                    # auto: the worker owns these sockets
    kinds = [type(m).__name__ for m in got]
    assert kinds == ['WorkerReady'] + ['Iterate'] * 6 + ['Shutdown']
    its = got[1:7]
    assert [m.i for m in its] == [1, 2, 3, 4, 5, 6] and [float(m.image[0, 0, 0]) for m in its] == [1, 2, 3, 4, 5, 6]
    assert all(m.image.dtype == F32 and m.image.shape == (2, 2, 3) and type(m.trace) is OrderedDict for m in its)


# --------------------------------------------------------------------- a peer that stops reading (libzmq's tracker semantics)
class NotDone(Exception):
    """pyzmq's zmq.NotDone: MessageTracker.wait(timeout) expired."""


class StallingSockets(FakeSockets):
    """libzmq releases a tracked zero-copy frame only once a connected peer has taken the bytes.  This outbound socket takes
    ``alive`` frames and then behaves like an app that died mid-stream: trackers never complete, ``wait(timeout)`` raises
    NotDone -- until the socket is closed (LINGER 0 frees the frames)."""
    def __init__(self, inbound, alive):
        super().__init__(inbound)
        self.alive, self.raw_sends, self.closed = alive, 0, threading.Event()
        self.stuck = []

    def send(self, data, copy=True, track=False):
        assert copy is False and track is True
        self.raw_sends += 1
        outer, taken = self, self.raw_sends <= self.alive
        if taken:
            outer.sent.append(pickle.loads(bytes(data)))
        else:
            outer.stuck.append(data)

        class Tracker:
            @property
            def done(self_inner):
                return taken or outer.closed.is_set()

            def wait(self_inner, timeout=None):
                if taken:
                    return
                if not outer.closed.wait(timeout):
                    raise NotDone()
        return Tracker()

    def send_pyobj(self, obj):
        if self.raw_sends > self.alive:          # the dead peer takes nothing more; libzmq queues small messages below its HWM
            return
        super().send_pyobj(obj)


@pytest.mark.parametrize('async_iterate', ['0', '1'])
def test_worker_exits_when_the_peer_disappears_with_a_frame_in_flight(async_iterate, monkeypatch):
    """with zero-copy iterates a dead app must not hang the worker -- the tracker wait is sliced, Shutdown /
    SIGHUP bound it by a grace period, queued frames are dropped and Worker.close() is reached (reference worker.py:362-363,
    429-431: Shutdown is queued and ctx.destroy(0) drops what the peer never took)."""
    import threading as th
    monkeypatch.setattr(worker_mod, 'SEND_SLICE_S', 0.02)
    monkeypatch.setattr(worker_mod, 'SHUTDOWN_GRACE_S', 0.3)
    img = np.zeros((4, 4, 3), np.uint8)
    socks = StallingSockets([messages.SetImages(None, img, img, img, True), messages.StartIteration()], alive=2)
    tr = FakeFramedTransfer(6)
    wk = worker_mod.Worker({'async_iterate': async_iterate, 'zero_copy_iterate': '1'}, sock_in=socks, sock_out=socks, transfer=tr)
    sender = wk.sock_out if async_iterate == '1' else None
    if sender is not None:
        sender.grace = 0.3
    done = th.Event()

    def body():
        try:
            wk.run()
        finally:
            wk.close()
            done.set()

    def hang_up():
        # what the reference's SIGHUP handler does (utils.py:187-190: KeyboardInterrupt in the worker's thread), delivered to the
        # thread this test runs the worker on; by then it sits in a sliced wait (the tracker's, or the full queue's)
        import ctypes
        ctypes.pythonapi.PyThreadState_SetAsyncExc(ctypes.c_ulong(t.ident), ctypes.py_object(KeyboardInterrupt))
    t = th.Thread(target=body, daemon=True)
    t0 = time.time()
    t.start()
    th.Timer(0.5, hang_up).start()
    assert done.wait(20.0), 'the worker hung on a frame the dead peer never took'
    assert time.time() - t0 < 10.0
    taken = [m for m in socks.sent if isinstance(m, messages.Iterate)]
    assert [m.i for m in taken] == [1, 2]                       # what the live peer took arrived intact and in order
    assert len(socks.stuck) >= 1                               # ... and at least one frame really was in flight when it died
    if async_iterate == '1':
        assert isinstance(wk.sock_out, StallingSockets) and not sender.thread.is_alive()     # the sender thread is gone


def test_async_sender_close_is_bounded_when_nothing_is_taken(monkeypatch):
    monkeypatch.setattr(worker_mod, 'SEND_SLICE_S', 0.02)
    socks = StallingSockets([], alive=0)
    sender = worker_mod.AsyncSender(socks, depth=2, grace=0.2)
    for k in range(3):                                           # one in flight on the sender thread + two queued
        sender.send_frame(memoryview(bytearray(pickle.dumps(k))))
        time.sleep(0.05)
    t0 = time.time()
    blocker = threading.Thread(target=lambda: sender.send_frame(memoryview(bytearray(pickle.dumps(9)))), daemon=True)
    blocker.start()                                              # the queue is full: this put waits (back-pressure) ...
    time.sleep(0.1)
    assert blocker.is_alive()
    sender.close()                                               # ... until close() gives up after the grace period
    blocker.join(5.0)
    assert not blocker.is_alive() and not sender.thread.is_alive()
    assert time.time() - t0 < 5.0 and sender.dropped >= 2 and socks.sent == []


def test_a_stored_sender_error_does_not_keep_the_context_alive(monkeypatch):
    """when the sender thread has stored an error, `send_pyobj(Shutdown())` in run()'s finally raises it again, and
    so does `sender.close()` -- Worker.close() must still reach ctx.destroy(0) (an undestroyed context blocks interpreter exit in zmq
    term with the default LINGER: the hang the bounded shutdown was written to remove), and run() must not die in its finally."""
    monkeypatch.setattr(worker_mod, 'SEND_SLICE_S', 0.02)
    monkeypatch.setattr(worker_mod, 'SHUTDOWN_GRACE_S', 0.3)

    class BrokenOut(StallingSockets):
        def send_pyobj(self, obj):
            if isinstance(obj, messages.Iterate):
                raise OSError('transport gone')
            super().send_pyobj(obj)

    class Ctx:
        destroyed = 0

        def destroy(self, linger):
            Ctx.destroyed += 1
    img = np.zeros((4, 4, 3), np.uint8)
    socks = BrokenOut([messages.SetImages(None, img, img, img, True), messages.StartIteration(), messages.Shutdown()], alive=100)
    wk = worker_mod.Worker({'async_iterate': '1', 'zero_copy_iterate': '0'}, sock_in=socks, sock_out=socks, transfer=FakeFramedTransfer(4))
    wk._ctx = Ctx()
    wk.run()                               # the stored OSError surfaces at a later send; the finally block must swallow it and go on
    with pytest.raises(OSError):
        # (whether close() re-raises the stored error is not the point -- the context is)
        wk.sock_out = worker_mod.AsyncSender(socks, depth=1, grace=0.2)
        wk.sock_out.error = OSError('transport gone')
        wk.close()
    assert Ctx.destroyed == 1
