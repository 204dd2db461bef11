"""A minimal PUSH / PULL pair speaking ZMTP 3.0 over TCP -- TEST INFRASTRUCTURE.

The reference's transport is pyzmq (worker.py:13,321-324: PULL bind ``worker_socket``, PUSH connect ``app_socket``; one message =
one ZMQ frame holding a pickle).  pyzmq is not installed in the build image nor on the GPU box, so the worker's own-socket code path
(``Worker(config)`` without injected sockets: sockets created from the config, ``send(frame, copy=False, track=True)`` for the
zero-copy iterates) would never execute in any record.  This module provides just enough of the ``zmq`` module's surface for that path
-- ``Context``, ``socket(PULL | PUSH)``, ``bind`` / ``bind_to_random_port`` / ``connect``, ``send_pyobj`` / ``recv_pyobj`` / ``send``,
``Poller``, ``NOBLOCK``, ``ZMQError`` -- on real TCP sockets with the published wire protocol (ZMTP 3.0, rfc.zeromq.org/spec/23: 64-byte
greeting, NULL security handshake with READY commands, short / long frames), so that a pyzmq peer would understand the bytes.
It is registered as ``sys.modules['zmq']`` by the loopback tests ONLY when the real pyzmq is absent; the product never imports it.
Interoperability with libzmq itself cannot be checked here (no libzmq); the framing follows the specification.
"""

import pickle
import select
import socket
import struct
import threading
import time

PULL, PUSH = 7, 8
NOBLOCK = 1
POLLIN = 1
DEFAULT_PROTOCOL = pickle.DEFAULT_PROTOCOL


class ZMQError(Exception):
    pass


class Again(ZMQError):
    pass


_NAMES = {PULL: b'PULL', PUSH: b'PUSH'}


def _greeting():
    return b'\xff' + b'\x00' * 8 + b'\x7f' + b'\x03\x00' + b'NULL'.ljust(20, b'\x00') + b'\x00' + b'\x00' * 31


def _ready(kind):
    name, value = b'Socket-Type', _NAMES[kind]
    body = b'\x05READY' + bytes([len(name)]) + name + struct.pack('>I', len(value)) + value
    return b'\x04' + bytes([len(body)]) + body


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ZMQError('peer closed the connection')
        buf += chunk
    return bytes(buf)


def _read_frame(sock):
    """(flags, payload) of the next frame."""
    flags = _recv_exact(sock, 1)[0]
    size = struct.unpack('>Q', _recv_exact(sock, 8))[0] if flags & 0x02 else _recv_exact(sock, 1)[0]
    data = bytearray(size)
    view, got = memoryview(data), 0
    while got < size:
        n = sock.recv_into(view[got:], size - got)
        if not n:
            raise ZMQError('peer closed the connection inside a frame')
        got += n
    return flags, data


def _handshake(sock, kind):
    sock.sendall(_greeting())
    peer = _recv_exact(sock, 64)
    if peer[0] != 0xff or peer[9] != 0x7f or peer[10] < 3 or not peer[12:32].startswith(b'NULL'):
        raise ZMQError('not a ZMTP 3.x NULL peer')
    sock.sendall(_ready(kind))
    flags, body = _read_frame(sock)
    if not flags & 0x04 or not bytes(body).startswith(b'\x05READY'):
        raise ZMQError('READY expected')


class MessageTracker:
    def __init__(self):
        self._done = threading.Event()

    def wait(self, timeout=None):
        self._done.wait(timeout)

    @property
    def done(self):
        return self._done.is_set()


class Socket:
    def __init__(self, kind):
        self.kind = kind
        self._listen = None
        self._conns = []
        self._lock = threading.Lock()
        self._closed = False
        self._endpoint = None
        self._rr = 0

    # ---- endpoints
    @staticmethod
    def _parse(addr):
        assert addr.startswith('tcp://'), addr
        host, port = addr[6:].rsplit(':', 1)
        return ('0.0.0.0' if host == '*' else host), int(port)

    def bind(self, addr):
        srv = socket.socket()
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind(self._parse(addr))
        srv.listen(8)
        srv.settimeout(0.05)                        # (accept() wakes up regularly so that close() really releases the port)
        self._listen = srv
        threading.Thread(target=self._accept_loop, daemon=True).start()

    def bind_to_random_port(self, addr):
        self.bind(addr + ':0')
        return self._listen.getsockname()[1]

    def _accept_loop(self):
        while not self._closed:
            try:
                conn, _ = self._listen.accept()
            except socket.timeout:
                continue
            except OSError:
                return
            try:
                conn.settimeout(None)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _handshake(conn, self.kind)
            except (OSError, ZMQError):
                conn.close()
                continue
            with self._lock:
                self._conns.append(conn)

    def connect(self, addr):
        self._endpoint = self._parse(addr)          # connected lazily, re-tried like libzmq does when the peer binds later

    def _ensure_connected(self, timeout=10.0):
        if self._conns or self._endpoint is None:
            return
        deadline = time.time() + timeout
        while True:
            try:
                conn = socket.create_connection(self._endpoint, timeout=2.0)
                conn.settimeout(None)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _handshake(conn, self.kind)
                with self._lock:
                    self._conns.append(conn)
                return
            except OSError:
                if time.time() > deadline:
                    raise ZMQError('cannot connect to %s:%d' % self._endpoint)
                time.sleep(0.05)

    # ---- PUSH
    def send(self, data, flags=0, copy=True, track=False):
        assert self.kind == PUSH
        self._ensure_connected()
        deadline = time.time() + 10.0
        while not self._conns:                      # a bound PUSH waits for its first peer
            if time.time() > deadline:
                raise ZMQError('no peer')
            time.sleep(0.01)
        view = memoryview(data).cast('B')
        n = len(view)
        head = (b'\x00' + bytes([n])) if n < 256 else (b'\x02' + struct.pack('>Q', n))
        with self._lock:
            conn = self._conns[self._rr % len(self._conns)]
            self._rr += 1
        conn.sendall(head)
        conn.sendall(view)                          # the kernel has every byte when this returns: the buffer may be reused
        if track:
            t = MessageTracker()
            t._done.set()
            return t
        return None

    def send_pyobj(self, obj, flags=0, protocol=DEFAULT_PROTOCOL):
        return self.send(pickle.dumps(obj, protocol), flags)

    # ---- PULL
    def _ready_conns(self, timeout):
        with self._lock:
            conns = list(self._conns)
        if not conns:
            time.sleep(min(timeout, 0.01) if timeout else 0)
            return []
        return select.select(conns, [], [], timeout)[0]

    def poll(self, timeout_ms=None, flags=POLLIN):
        deadline = None if timeout_ms is None else time.time() + timeout_ms / 1000.0
        while True:
            if self._ready_conns(0.01):
                return POLLIN
            if deadline is not None and time.time() >= deadline:
                return 0

    def recv(self, flags=0):
        assert self.kind == PULL
        while True:
            ready = self._ready_conns(0 if flags & NOBLOCK else 0.05)
            for conn in ready:
                try:
                    fl, data = _read_frame(conn)
                except ZMQError:
                    with self._lock:
                        if conn in self._conns:
                            self._conns.remove(conn)
                    conn.close()
                    continue
                if fl & 0x04:
                    continue                        # a command (e.g. a heartbeat): not a message
                return data
            if flags & NOBLOCK:
                raise Again('Resource temporarily unavailable')

    def recv_pyobj(self, flags=0):
        return pickle.loads(self.recv(flags))

    def close(self, linger=None):
        self._closed = True
        if self._listen is not None:
            time.sleep(0.06)                        # the accept loop has seen _closed and left accept()
            try:
                self._listen.close()
            except OSError:
                pass
        with self._lock:
            for conn in self._conns:
                try:
                    conn.close()
                except OSError:
                    pass
            self._conns = []


class Context:
    def __init__(self):
        self._socks = []

    def socket(self, kind):
        s = Socket(kind)
        self._socks.append(s)
        return s

    def destroy(self, linger=None):
        for s in self._socks:
            s.close(linger)
        self._socks = []

    term = destroy


class Poller:
    def __init__(self):
        self._socks = []

    def register(self, sock, flags=POLLIN):
        self._socks.append(sock)

    def poll(self, timeout_ms=None):
        deadline = None if timeout_ms is None else time.time() + timeout_ms / 1000.0
        while True:
            hits = [(s, POLLIN) for s in self._socks if s._ready_conns(0.005)]
            if hits:
                return hits
            if deadline is not None and time.time() >= deadline:
                return []
