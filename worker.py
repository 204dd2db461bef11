#!/usr/bin/env python3
"""The style-transfer worker process, MI355X edition.

Drop-in for the reference's ``worker.py`` (launched by ``app.py`` as ``worker.py [config] [-d...]``,
reference app.py:38,341): same config keys, same sockets (PULL bind ``worker_socket``, PUSH connect
``app_socket``), same message protocol and ordering (``WorkerReady`` first, one ``Iterate`` per step,
``GetImages`` when it cannot iterate, ``Shutdown`` always last), exit code 2 when the compute backend
is missing (reference worker.py:51-53).  The per-iteration work runs on the GPU through
``style_transfer2_amd``; there is no CPU fallback.
"""

import argparse
import configparser
import logging
import os
from pathlib import Path
import queue
import signal
import sys
import threading
import time

MODULE_DIR = Path(__file__).parent.resolve()
sys.path.insert(0, str(MODULE_DIR))

from messages import (GetImages, Iterate, PauseIteration, SetImages, SetOptimizer, SetWeights,  # noqa: E402
                      Shutdown, StartIteration, WorkerReady)
import messages  # noqa: E402

logger = logging.getLogger('worker')

BACKEND_MSG = '''
Error: the MI355X compute backend is unavailable (%s).
Build it with `python -c "import __graft_entry__ as g; g.build()"` and check `gpu` in the config.'''


# ---------------------------------------------------------------------------------------------------
# config / logging / signals (reference utils.py:110-127, 172-190)
# ---------------------------------------------------------------------------------------------------
def parse_args(desc=''):
    """``worker.py [config] [-d ...]`` (reference utils.py:110-117): an optional config path and a repeatable debug flag."""
    cli = argparse.ArgumentParser(description=desc)
    cli.add_argument('config', nargs='?', default=None, help='the config file')
    cli.add_argument('-d', '--debug', action='count', default=0, help='debug (repeat for message creation sites)')
    return cli.parse_args()


def read_config(args):
    cp = configparser.ConfigParser()
    files = [str(MODULE_DIR / 'config.ini'), str(MODULE_DIR / 'config_non_git.ini')]
    if args.config:
        files.append(args.config)
    cp.read(files)
    return cp['DEFAULT']


def setup_logging(debug=0):
    logging.basicConfig(level=logging.DEBUG if debug else logging.INFO, datefmt='%H:%M:%S',
                        format='%(asctime)s.%(msecs)03d %(process)d %(name)s %(levelname)s: %(message)s')
    if debug > 1:
        messages.Message.debug = True
    logging.captureWarnings(True)


def setup_signals():
    def on_hup(*_):
        raise KeyboardInterrupt()
    signal.signal(signal.SIGHUP, on_hup)


# ---------------------------------------------------------------------------------------------------
def build_transfer(config):
    """Model + StyleTransfer for this worker (reference worker.py:326-332).  Exits with code 2 when the
    HIP library, the GPU or the weights are missing."""
    try:
        import style_transfer2_amd as st2
        from style_transfer2_amd import weights as st2_weights
        gpu = config.getint('gpu', fallback=0)
        # the network definition (reference config.ini:28, worker.py:58-61): the stock VGG19 is built in, a prototxt that exists
        # is read (and refused loudly if it asks for anything the engine does not run)
        proto_path = MODULE_DIR / config.get('prototxt', 'models/vgg19.prototxt')
        topology = st2.VGG19_TOPOLOGY
        if proto_path.exists():
            from style_transfer2_amd import prototxt
            topology = prototxt.read(str(proto_path))
        weights_path = MODULE_DIR / config.get('caffemodel', 'models/vgg19.npz')
        if weights_path.suffix == '.npz' and weights_path.exists():
            params = st2_weights.load_npz(str(weights_path), topology)
        elif weights_path.suffix == '.caffemodel' and weights_path.exists():
            from style_transfer2_amd import caffemodel
            params = caffemodel.vgg_params(caffemodel.read_caffemodel(str(weights_path)), topology,
                                           bgr_to_rgb=config.getboolean('caffemodel_is_bgr', fallback=False))
        elif config.get('weights', '') == 'synthetic':
            logger.warning('Using seeded synthetic weights (config: weights = synthetic).')
            params = st2_weights.he_normal(topology, seed=0)
        else:
            raise st2.HipUnavailable('weights file %s not found (.npz or .caffemodel; or set weights = synthetic)'
                                     % weights_path)
        model = st2.HipModel(params, topology=None if topology == st2.VGG19_TOPOLOGY else topology, device=max(gpu, 0),
                             precision=config.get('precision', 'fp32'))
        return st2.StyleTransfer(model)
    except Exception as err:  # HipUnavailable, StError, OSError ...
        print(BACKEND_MSG % err, file=sys.stderr)
        sys.exit(2)


class _Frame:
    __slots__ = ('buf',)

    def __init__(self, buf):
        self.buf = buf


class SendAborted(RuntimeError):
    """A zero-copy frame was given up: the peer never took it and the worker is shutting down (or the wait timed out)."""


SEND_SLICE_S = 0.25          # a tracked send is waited for in slices of this length, so that a stop request is seen
SHUTDOWN_GRACE_S = 5.0       # how long a shutting-down worker keeps trying to hand its last messages to a peer that is not reading


def send_frame(sock, buf, stop=None, timeout=None):
    """One already-pickled message, straight from pinned memory: no copy on this side (pyzmq ``send(copy=False)``).  The
    buffer belongs to the engine and is rewritten a few iterations later, so the call returns only once the transport no
    longer needs it (``track=True``: libzmq has handed the bytes to a connected peer).  libzmq releases a tracked frame only
    when a peer takes it -- if the app died mid-stream that is never -- so the wait is taken in bounded slices
    (``MessageTracker.wait(timeout)`` raises ``zmq.NotDone`` on expiry) and ends with ``SendAborted`` when ``stop`` is set
    or ``timeout`` seconds have passed; the caller then closes the socket with LINGER 0, which frees the frame."""
    tracker = sock.send(buf, copy=False, track=True)
    if tracker is None or not hasattr(tracker, 'wait'):
        return
    deadline = None if timeout is None else time.monotonic() + timeout
    while True:
        try:
            tracker.wait(SEND_SLICE_S)
        except TypeError:                 # a tracker without a timeout argument (test doubles): one unbounded wait
            tracker.wait()
            return
        except Exception as err:          # zmq.NotDone: not sent yet
            if type(err).__name__ != 'NotDone':
                raise
        if getattr(tracker, 'done', True):
            return
        if (stop is not None and stop.is_set()) or (deadline is not None and time.monotonic() > deadline):
            raise SendAborted('the peer did not take an Iterate frame')


class AsyncSender:
    """Owns the outbound socket on its own thread: pickling + sending an ``Iterate`` (12.6 MB at 1024^2)
    overlaps the next iteration's GPU work instead of stalling it.  Order is
    preserved (one FIFO), every message still goes out exactly once, and the queue is bounded so the
    worker never runs more than ``depth`` iterates ahead of what the app has been sent.

    While the worker iterates, a peer that reads slowly simply holds the worker back (as libzmq's high-water mark does in the
    reference).  Once ``begin_shutdown`` has been called nothing waits longer than ``grace`` seconds: messages that a dead
    peer will never take are dropped, the thread ends, and the caller can destroy the context (reference worker.py:362-363,
    429-431: ``Shutdown`` is queued and ``ctx.destroy(0)`` drops what could not be delivered)."""

    def __init__(self, sock, depth=2, grace=SHUTDOWN_GRACE_S):
        self.sock = sock
        self.q = queue.Queue(maxsize=depth)
        self.error = None
        self.grace = grace
        self.stop = threading.Event()          # set: give up on whatever is in flight or queued
        self.deadline = None                   # set by begin_shutdown
        self.dropped = 0
        self.thread = threading.Thread(target=self._run, name='iterate-sender', daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            msg = self.q.get()
            if msg is None:
                return
            if self.stop.is_set():
                self.dropped += 1
                continue
            try:
                if isinstance(msg, _Frame):
                    send_frame(self.sock, msg.buf, stop=self.stop)
                else:
                    self.sock.send_pyobj(msg)
            except SendAborted:
                self.dropped += 1
            except Exception as err:      # surfaced on the worker thread at the next send
                self.error = err

    def _put(self, item):
        """Queue one item; False if the sender has been stopped or the shutdown grace period ran out first."""
        while True:
            if self.stop.is_set():
                return False
            try:
                self.q.put(item, timeout=SEND_SLICE_S)
                return True
            except queue.Full:
                if self.deadline is not None and time.monotonic() > self.deadline:
                    self.abort()
                    return False

    def send_pyobj(self, msg):
        if self.error is not None:
            raise self.error
        if not self._put(msg):
            self.dropped += 1

    def send_frame(self, buf):
        """A finished pickle (iterate_frame.py) that lives in the engine's rotating pinned buffers: sent as it is."""
        self.send_pyobj(_Frame(buf))

    def begin_shutdown(self):
        """From now on nothing blocks for longer than the grace period."""
        if self.deadline is None:
            self.deadline = time.monotonic() + self.grace

    def abort(self):
        """Give up on everything queued or in flight (the peer is not reading); the thread then drains its queue and ends."""
        self.stop.set()

    def close(self):
        """Flush everything queued -- for at most the grace period -- then stop the thread."""
        self.begin_shutdown()
        queued = self._put(None)              # False: the grace period ran out with the queue still full
        if queued:
            self.thread.join(max(0.0, self.deadline - time.monotonic()) + SEND_SLICE_S)
        if self.thread.is_alive():
            logger.warning('the peer is not reading: dropping the messages still queued')
            self.abort()                      # the wait on the frame in flight ends within one slice; queued messages are dropped
            give_up = time.monotonic() + 8 * SEND_SLICE_S
            while not queued and time.monotonic() < give_up:
                try:
                    self.q.put_nowait(None)
                    queued = True
                except queue.Full:
                    time.sleep(0.02)
            self.thread.join(max(0.0, give_up - time.monotonic()))
        if self.error is not None:
            raise self.error


class Worker:
    """Message loop (reference worker.py:318-409).  ``sock_in``/``sock_out`` may be injected (tests);
    otherwise pyzmq PULL/PUSH sockets are created from the config."""

    def __init__(self, config, sock_in=None, sock_out=None, transfer=None):
        self._ctx = None
        if sock_in is None or sock_out is None:
            import zmq
            self._ctx = zmq.Context()
            sock_in = self._ctx.socket(zmq.PULL)
            sock_out = self._ctx.socket(zmq.PUSH)
            sock_in.bind(config['worker_socket'])
            sock_out.connect(config['app_socket'])
            self._again = zmq.ZMQError
            self._noblock = zmq.NOBLOCK
        else:
            self._again = getattr(sock_in, 'Again', BlockingIOError)
            self._noblock = 1
        self.sock_in = sock_in
        self._raw_out = self.sock_out = sock_out
        self.run_should_stop = False
        self._shutting_down = False
        try:
            # the model first (it may sys.exit(2): reference worker.py:51-53), the sender thread only once it exists
            self.transfer = transfer if transfer is not None else build_transfer(config)
            async_send = str(config.get('async_iterate', '1')).lower() not in ('0', 'false', 'no')
            if async_send:
                self.sock_out = AsyncSender(sock_out)
            # pipelined iterations (backends that offer step_begin / step_end): iteration k + 1 is queued on the GPU before
            # iterate k is collected and sent.  What goes out, and in which order, is unchanged: every pending iterate is sent
            # before a received message is acted on.
            self.pipelined = (hasattr(self.transfer, 'step_begin') and
                              str(config.get('pipeline_iterate', '1')).lower() not in ('0', 'false', 'no'))
            # Zero-copy iterates: the GPU copies each iterate into the payload slot of a pre-formatted pickle in pinned memory
            # (iterate_frame.py) and the transport sends that buffer as it is -- no 12.6 MB (50 MB at 2048^2) host copy or pickling.
            # Only for the pyzmq sockets this worker created itself (their send(copy=False, track=True) is what bounds the
            # buffer's use) or when the config asks for it; an injected socket gets owned copies through send_pyobj.
            want = str(config.get('zero_copy_iterate', 'auto')).lower()
            self.zero_copy = (self.pipelined and hasattr(self.transfer, 'enable_iterate_frames') and
                              (want in ('1', 'true', 'yes') or (want == 'auto' and self._ctx is not None)))
            if self.zero_copy:
                self.transfer.enable_iterate_frames()
            self.sock_out.send_pyobj(WorkerReady(layers=self.transfer.model.layers()))
        except BaseException:
            self.close()            # the reference always reaches ctx.destroy(0) (worker.py:429-431)
            raise

    def close(self):
        # ctx.destroy(0) must be reached whatever the sender thread has stored (an error of its own is raised again by its close()):
        # an undestroyed context blocks the interpreter's exit in zmq term with the default LINGER
        try:
            if isinstance(self.sock_out, AsyncSender):
                sender, self.sock_out = self.sock_out, self._raw_out
                sender.close()
        finally:
            if self._ctx is not None:
                self._ctx.destroy(0)

    def _collect_and_send(self):
        """The oldest iteration in flight: wait for it, send its Iterate."""
        tr = self.transfer
        if self.zero_copy:
            _, _, _, frame = tr.step_end(frame=True)
            if isinstance(self.sock_out, AsyncSender):
                self.sock_out.send_frame(frame)
            else:
                send_frame(self.sock_out, frame, timeout=SHUTDOWN_GRACE_S if self._shutting_down else None)
        else:
            image, trace, index = tr.step_end()
            self.sock_out.send_pyobj(Iterate(image, index, trace))

    def _flush_pending(self):
        """Collect and send every iteration that was begun (pipelined mode), oldest first."""
        while getattr(self.transfer, 'steps_pending', 0):
            self._collect_and_send()

    def run(self):
        try:
            while not self.run_should_stop:
                if self.transfer.is_running:
                    self._drain_then_step()
                else:
                    self._flush_pending()
                    if self.process_message(self.sock_in.recv_pyobj()):
                        break
        except KeyboardInterrupt:
            pass
        finally:
            self._shutting_down = True
            if isinstance(self.sock_out, AsyncSender):       # from here on a peer that is not reading cannot hold the exit up
                self.sock_out.begin_shutdown()
            try:
                self._flush_pending()
            except Exception:                 # the reference always reaches Shutdown (worker.py:396-398)
                logger.exception('could not collect the iterations in flight')
            try:
                self.sock_out.send_pyobj(Shutdown())
            except Exception:                 # an error the sender thread stored earlier is raised by this send: the peer is gone or the
                logger.exception('Shutdown could not be queued')      # transport failed -- close() below still runs, the exit stays bounded
            if isinstance(self.sock_out, AsyncSender):       # Shutdown is the last thing on the wire
                sender, self.sock_out = self.sock_out, self._raw_out
                try:
                    sender.close()
                except Exception:
                    logger.exception('sender thread ended with an error')

    def _drain_then_step(self):
        """Handle everything queued without blocking, then do exactly one iteration."""
        try:
            while True:
                msg = self.sock_in.recv_pyobj(self._noblock)
                self._flush_pending()             # whatever was computed before the message arrived goes out first
                if self.process_message(msg):
                    self.run_should_stop = True
                    return
        except self._again:
            pass
        if not self.transfer.is_running:
            return
        if not self.transfer.check_consistency():
            self._flush_pending()
            self.sock_out.send_pyobj(GetImages())
        elif self.pipelined:
            self.transfer.step_begin()            # queue iteration k + 1 ...
            if self.transfer.steps_pending > 1:   # ... then collect and send iterate k while the GPU works on it
                self._collect_and_send()
        else:
            image, trace = self.transfer.step()
            self.sock_out.send_pyobj(Iterate(image, self.transfer.t, trace))

    # what each inbound message does (reference worker.py:366-409); Shutdown is the only one that ends the loop
    def _on_set_images(self, msg):
        tr = self.transfer
        for value, assign, resample in ((msg.input_image, tr.set_input, tr.resample_input),
                                        (msg.content_image, tr.set_content, tr.resample_content),
                                        (msg.style_image, tr.set_style, None)):
            if value is None:
                continue
            if not isinstance(value, int):                    # an image
                assign(value)
            elif value == SetImages.RESAMPLE and resample is not None:
                resample(msg.size)
        if msg.reset_state:
            tr.reset()

    def _on_set_optimizer(self, msg):
        tr = self.transfer
        tr.optimizer_cls = SetOptimizer.classes[msg.optimizer]
        tr.set_step_size(msg.step_size)
        if not isinstance(tr.optimizer, tr.optimizer_cls):
            tr.reset()

    def _on_set_weights(self, msg):
        self.transfer.set_weights(msg.weights, msg.params)

    def _on_start(self, msg):
        if not self.transfer.start():
            self.sock_out.send_pyobj(GetImages())

    def _on_pause(self, msg):
        self.transfer.pause()

    HANDLERS = ((SetImages, _on_set_images), (SetOptimizer, _on_set_optimizer), (SetWeights, _on_set_weights),
                (StartIteration, _on_start), (PauseIteration, _on_pause))

    def process_message(self, msg):
        """Returns True when the worker should shut down."""
        if isinstance(msg, Shutdown):
            return True
        for kind, handler in self.HANDLERS:
            if isinstance(msg, kind):
                handler(self, msg)
                return False
        logger.error('Invalid message received over ZeroMQ.')
        return False


def main():
    args = parse_args(__doc__)
    config = read_config(args)
    debug = (args.debug or 0) + config.getint('debug', 0)
    setup_logging(debug)
    setup_signals()
    worker = None
    try:
        worker = Worker(config)
        worker.run()
    except ImportError as err:          # pyzmq missing
        print(BACKEND_MSG % err, file=sys.stderr)
        sys.exit(2)
    finally:
        logger.info('Shutting down worker process.')
        if worker is not None:
            worker.close()


if __name__ == '__main__':
    main()
