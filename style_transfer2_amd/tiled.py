"""Driver of the tile-sharded single-image mode (see tiling.py for the design).

``TiledTransfer`` sequences the phases of one Adam iteration on every rank and performs the communication
between them through ``Comm`` (torch.distributed: RCCL on the GPUs, gloo in the CPU tests).  The compute
of each phase is delegated to a tile backend:

    backend.forward_partials()      -> 1-D float32 tensor  [per-layer scalar sums | raw Gram sums]   (all-reduced)
    backend.losses_need_style_norm()-> 1-D tensor of sum S^2 per style layer, or None               (all-reduced)
    backend.finish_losses()
    backend.backward()              -> (3, wh, ww) window gradient tensor                            (overlap-add)
    backend.update(ring)            -> 1-D tensor of image-space partial sums                        (all-reduced)
    backend.finish_trace()          -> trace values (same layout as the single-GPU engine)
    backend.x_next()                -> (3, wh, ww) tensor holding the updated tile (aprons stale)    (refresh)
    backend.swap()

The HIP backend (``HipTileBackend``) keeps every tensor on the GPU; tests run the same driver over a numpy
backend built on the CPU oracle.
"""

import numpy as np

from .tiling import Rect


class _LazyTorch:
    """Only the phase-by-phase driver (TiledTransfer / Comm) uses torch; FusedTiledTransfer (st_tile_step over RCCL) does not."""
    def __getattr__(self, name):
        import torch as _torch
        return getattr(_torch, name)


torch = _LazyTorch()


class Comm:
    """all-reduce and rectangle exchange over torch.distributed; a no-op group for world == 1.
    Device tensors are staged through the host when the backend is gloo (tests)."""

    def __init__(self, dist=None, rank=0, world=1):
        self.dist, self.rank, self.world = dist, rank, world
        self.staged = dist is not None and dist.get_backend() == 'gloo'

    def all_reduce(self, t):
        if self.dist is None or self.world == 1 or t is None or t.numel() == 0:
            return t
        if self.staged and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        else:
            self.dist.all_reduce(t)
        return t

    def exchange(self, sends, recvs):
        """sends: [(peer, tensor)], recvs: [(peer, tensor)] in matching plan order; tensors contiguous."""
        if self.dist is None or self.world == 1:
            return
        ops, stage = [], []
        for i, (peer, t) in enumerate(sends):
            buf = t.cpu() if (self.staged and t.is_cuda) else t
            ops.append(self.dist.P2POp(self.dist.isend, buf, peer))
        for i, (peer, t) in enumerate(recvs):
            if self.staged and t.is_cuda:
                buf = torch.empty(t.shape, dtype=t.dtype)
                stage.append((t, buf))
            else:
                buf = t
            ops.append(self.dist.P2POp(self.dist.irecv, buf, peer))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for t, buf in stage:
            t.copy_(buf)


def _local(rect, origin):
    return slice(rect.y0 - origin.y0, rect.y1 - origin.y0), slice(rect.x0 - origin.x0, rect.x1 - origin.x0)


class TiledTransfer:
    """optimizer='adam': the backend's fused image pass + Adam update (one evaluation per step).
    optimizer='lbfgs': LBFGSOptimizer (optimizers.py:49-125) over the tile-sharded image: every rank keeps its tile of x, of the
    gradient and of the <= 10 curvature pairs; each utils.dot of the two-loop recursion is a per-rank partial sum + one scalar
    all-reduce, every axpy is local.  The backend supplies gradient / grad_tile / vdot / vaxpy / vscale / vcopy / apply_step."""

    N_CORR = 10                                     # optimizers.py:52

    def __init__(self, grid, rank, backend, comm, optimizer='adam', step_size=None):
        self.grid, self.rank, self.backend, self.comm = grid, rank, backend, comm
        self.window, self.tile = grid.windows[rank], grid.tiles[rank]
        if optimizer not in ('adam', 'lbfgs'):
            raise ValueError('optimizer must be adam or lbfgs')
        self.optimizer = optimizer
        self.step_size = np.float32(1 if step_size is None else step_size)      # L-BFGS only; Adam's lives in the backend
        self.sk, self.yk, self.syk = [], [], []
        self.lb_grad = None
        self.t = 0
        self._refresh = grid.apron_refresh_plan()
        self._overlap = grid.grad_overlap_plan()
        self._ring = grid.ring_plan()

    # -- communication phases ------------------------------------------------------------------------------
    # Everything one neighbour gets in a phase travels as ONE message: its rectangles are packed into a contiguous buffer
    # (one kernel per neighbour on the HIP backend, `backend.strips`) and unpacked the same way on the other side.
    MAX_RECTS = 12                              # st2_kernels.h kMaxStripRects

    def _pack(self, t, rects):
        n = sum(t.shape[0] * r[2] * r[3] for r in rects)
        if t.is_cuda and hasattr(self.backend, 'strips') and len(rects) <= self.MAX_RECTS:
            buf = torch.empty(n, dtype=t.dtype, device=t.device)
            self.backend.strips(t, rects, buf, 0)
            return buf
        return torch.cat([t[:, y:y + h, x:x + w].reshape(-1) for y, x, h, w in rects])

    def _unpack(self, t, rects, buf, add):
        if t.is_cuda and hasattr(self.backend, 'strips') and len(rects) <= self.MAX_RECTS:
            self.backend.strips(t, rects, buf, 2 if add else 1)
            return
        pos = 0
        for y, x, h, w in rects:
            n = t.shape[0] * h * w
            piece = buf[pos:pos + n].reshape(t.shape[0], h, w)
            if add:
                t[:, y:y + h, x:x + w] += piece
            else:
                t[:, y:y + h, x:x + w] = piece
            pos += n

    def _exchange(self, src, send_rects, dst, recv_rects, add):
        """send_rects / recv_rects: {peer: [(y0, x0, h, w), ...]} in plan order (local coordinates of src / dst)."""
        sends = [(peer, self._pack(src, rects)) for peer, rects in sorted(send_rects.items())]
        recvs = [(peer, torch.empty(sum(dst.shape[0] * r[2] * r[3] for r in rects), dtype=dst.dtype, device=dst.device))
                 for peer, rects in sorted(recv_rects.items())]
        self.comm.exchange(sends, recvs)
        for (peer, buf) in recvs:                   # ascending peer, plan order inside: deterministic sums
            self._unpack(dst, recv_rects[peer], buf, add)

    @staticmethod
    def _lrect(rect, origin):
        return (rect.y0 - origin.y0, rect.x0 - origin.x0, rect.y1 - rect.y0, rect.x1 - rect.x0)

    def refresh_aprons(self, x):
        """x: (3, wh, ww) window tensor whose TILE part is current: fill the apron from the owners."""
        sends, recvs = {}, {}
        for src, dst, rect in self._refresh:
            if src == self.rank:
                sends.setdefault(dst, []).append(self._lrect(rect, self.window))
            elif dst == self.rank:
                recvs.setdefault(src, []).append(self._lrect(rect, self.window))
        self._exchange(x, sends, x, recvs, add=False)

    def overlap_add(self, g):
        """g: (3, wh, ww) window gradient: add the neighbours' contributions to MY tile pixels."""
        sends, recvs = {}, {}
        for src, dst, rect in self._overlap:
            if src == self.rank:
                sends.setdefault(dst, []).append(self._lrect(rect, self.window))
            elif dst == self.rank:
                recvs.setdefault(src, []).append(self._lrect(rect, self.window))
        self._exchange(g, sends, g, recvs, add=True)

    def gather_ring(self, x):
        """(3, th+2, tw+2) tensor: the tile's 1-px neighbourhood under the image's periodic wrap."""
        th, tw = self.tile.y1 - self.tile.y0, self.tile.x1 - self.tile.x0
        ring = torch.zeros((3, th + 2, tw + 2), dtype=x.dtype, device=x.device)
        sends, recvs = {}, {}
        for dst, items in enumerate(self._ring):
            for src, rect, ry, rx in items:
                h, w = rect.y1 - rect.y0, rect.x1 - rect.x0
                if src == self.rank:
                    ys, xs = _local(rect, self.window)      # my own tile pixels, in my window
                    if dst == self.rank:
                        ring[:, ry:ry + h, rx:rx + w] = x[:, ys, xs]
                    else:
                        sends.setdefault(dst, []).append(self._lrect(rect, self.window))
                elif dst == self.rank:
                    recvs.setdefault(src, []).append((ry, rx, h, w))
        self._exchange(x, sends, ring, recvs, add=False)
        return ring

    # -- L-BFGS over the sharded image ----------------------------------------------------------------------
    def _evaluate(self):
        """opfunc at the current image: the trace values; the backend's grad_tile() then holds this rank's part of the gradient."""
        b = self.backend
        self.comm.all_reduce(b.forward_partials())
        extra = b.losses_need_style_norm()
        if extra is not None:
            self.comm.all_reduce(extra)
        b.finish_losses()
        g = b.backward()
        self.overlap_add(g)
        ring = self.gather_ring(b.x_cur())
        self.comm.all_reduce(b.gradient(ring))
        return b.finish_trace()

    def _dot(self, a, c):
        """utils.dot over the whole image: partial sums of the ranks, all-reduced.  Returned as a python float holding the
        float32 sum, as scipy's sdot returns it (utils.py:29-36): the scalar arithmetic below is then done in double and rounded
        once where it meets an array, exactly as the reference's python floats are (and as lbfgs.hip does on one GPU)."""
        t = self.backend.vdot(a, c)
        self.comm.all_reduce(t)
        return float(np.float32(t.reshape(-1)[0].item()))

    def _inv_hv(self, grad):
        """optimizers.py:89-108, statement by statement."""
        b = self.backend
        p = b.vcopy(grad)
        alphas = []
        for s, y, sy in zip(reversed(self.sk), reversed(self.yk), reversed(self.syk)):
            alphas.append(self._dot(s, p) / sy)                 # double / double
            b.vaxpy(-alphas[-1], y, p)                          # saxpy: the coefficient is rounded to float32 here
        if self.sk:
            sy, y = self.syk[-1], self.yk[-1]
            b.vscale(sy / self._dot(y, y), p)                   # p *= python float: rounded to float32, one multiply
        else:       # no curvature information: a unit-RMS direction (p.size is the WHOLE image's)
            n = 3 * self.grid.gH * self.grid.gW
            b.vdiv(float(np.sqrt(self._dot(p, p) / n)), p)      # p /= np.float64: divided in double, rounded once
        for s, y, sy, alpha in zip(self.sk, self.yk, self.syk, reversed(alphas)):
            beta = self._dot(y, p) / sy
            b.vaxpy(alpha - beta, s, p)
        return p

    def _lbfgs_step(self):
        b = self.backend
        if self.lb_grad is None:                    # optimizers.py:64-65
            self._evaluate()
            self.lb_grad = b.grad_tile()
        s = self._inv_hv(self.lb_grad)              # s = -step_size * inv_hv(grad)            :68
        b.vscale(-float(self.step_size), s)
        b.apply_step(s)                             # x += s                                   :69
        self.refresh_aprons(b.x_next())
        b.swap()
        values = self._evaluate()                   # loss, grad = opfunc(x)                   :72
        g1 = b.grad_tile()
        y = b.vcopy(g1)
        b.vaxpy(-1.0, self.lb_grad, y)              # y = grad - self.grad                     :73
        sy = self._dot(s, y)                        # store_curvature_pair                     :79-87
        if sy > 1e-10:
            self.sk.append(s); self.yk.append(y); self.syk.append(sy)
        if len(self.sk) > self.N_CORR:
            self.sk, self.yk, self.syk = self.sk[1:], self.yk[1:], self.syk[1:]
        self.lb_grad = g1
        return values

    # -- one iteration ----------------------------------------------------------------------------------------
    def step(self):
        b = self.backend
        self.t += 1
        if self.optimizer == 'lbfgs':
            return self._lbfgs_step()
        self.comm.all_reduce(b.forward_partials())
        extra = b.losses_need_style_norm()
        if extra is not None:
            self.comm.all_reduce(extra)
        b.finish_losses()
        g = b.backward()
        self.overlap_add(g)
        ring = self.gather_ring(b.x_cur())
        self.comm.all_reduce(b.update(ring))
        values = b.finish_trace()
        self.refresh_aprons(b.x_next())
        b.swap()
        return values

    def tile_image(self):
        """This rank's tile of the current iterate as (th, tw, 3) float32 RGB (deprocessed)."""
        x = self.backend.x_cur()
        ys, xs = _local(self.tile, self.window)
        mean = torch.tensor((123.68, 116.779, 103.939), dtype=x.dtype, device=x.device).reshape(3, 1, 1)
        return (x[:, ys, xs] + mean).permute(1, 2, 0).contiguous().cpu().numpy()


PLAN_OVERLAP, PLAN_RING, PLAN_REFRESH = 0, 1, 2          # st2.h ST_TILE_PLAN_*


def fused_plans(grid, rank):
    """The three strip exchanges of one iteration as st_tile_plan wants them: {phase: {peer: (send rects, recv rects)}}, rects
    (y0, x0, h, w) in the coordinates of the tensor packed (the window) / unpacked into (the window; the ring for PLAN_RING).
    Same plans, same order, as TiledTransfer.overlap_add / gather_ring / refresh_aprons build per step."""
    window, tile = grid.windows[rank], grid.tiles[rank]

    def lrect(rect, origin):
        return (rect.y0 - origin.y0, rect.x0 - origin.x0, rect.y1 - rect.y0, rect.x1 - rect.x0)
    plans = {PLAN_OVERLAP: {}, PLAN_RING: {}, PLAN_REFRESH: {}}
    for phase, plan in ((PLAN_OVERLAP, grid.grad_overlap_plan()), (PLAN_REFRESH, grid.apron_refresh_plan())):
        for src, dst, rect in plan:
            if src == rank:
                plans[phase].setdefault(dst, ([], []))[0].append(lrect(rect, window))
            elif dst == rank:
                plans[phase].setdefault(src, ([], []))[1].append(lrect(rect, window))
    for dst, items in enumerate(grid.ring_plan()):
        for src, rect, ry, rx in items:
            h, w = rect.y1 - rect.y0, rect.x1 - rect.x0
            if src == rank:
                plans[PLAN_RING].setdefault(dst, ([], []))[0].append(lrect(rect, window))
                if dst == rank:                                     # the periodic wrap lands on this rank's own tile: a local copy
                    plans[PLAN_RING][dst][1].append((ry, rx, h, w))
            elif dst == rank:
                plans[PLAN_RING].setdefault(src, ([], []))[1].append((ry, rx, h, w))
    return plans


class FusedTiledTransfer:
    """The tile-sharded iteration -- Adam, or L-BFGS (HipTileBackend(optimizer='lbfgs'): the Gram form, ONE all-reduce of the new inner
    products per step) -- with the communication INSIDE the engine (st_tile_step): the backend owns an RCCL communicator (or, in tests,
    a host-staged transport), the exchange plans are handed over once, and a step is one call -- every compute phase, all-reduce and
    strip exchange is enqueued on the engine's stream, the host synchronises once per iteration for the trace.
    TiledTransfer (above) remains as the backend-agnostic statement of the plan: the CPU / gloo tests run it over the oracle."""

    def __init__(self, grid, rank, backend):
        self.grid, self.rank, self.backend = grid, rank, backend
        self.tile = grid.tiles[rank]
        self.t = 0
        for phase, peers in fused_plans(grid, rank).items():
            backend.set_plan(phase, peers)

    def step(self):
        self.t += 1
        return self.backend.step_fused()

    def step_async(self):
        """One iteration with nothing read back (no trace, no host synchronisation): the device loop of a headless job."""
        self.t += 1
        self.backend.step_fused_async()

    def tile_image(self):
        return self.backend.tile_image()


class InProcessFabric:
    """Transport between ranks that are THREADS of one process -- every rank an engine context on the same GPU: the way one GPU runs an
    image no single engine can hold (8192^2: eight windows of 25 GB, time-sliced), and the way the tests run all eight ranks of the
    2 x 4 layout on a one-GPU box (RCCL refuses two ranks on one device; a box admits six processes on its card).
    All-reduce = sum in rank order behind a barrier; exchange = one FIFO mailbox per (source, destination).  Payloads are numpy arrays
    (HipTileBackend.comm_init_host: staged through the host) or torch device tensors (comm_init_local: device-to-device copies; the
    sender returns only when its messages have been consumed, its pack buffers are reused by the next phase)."""

    def __init__(self, world, timeout=120.0):
        import threading
        self.world, self.timeout = world, timeout
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.cond = threading.Condition()
        self.mail = {}
        self.pending = [0] * world                 # messages of rank r not yet consumed (device mode)
        self.reduces = self.messages = 0
        self.aborted = False                       # a rank failed: every wait of the others ends at once (abort())

    def abort(self):
        """A rank raised: wake every rank that waits for it (barrier and mailboxes) instead of letting them run into the timeout."""
        with self.cond:
            self.aborted = True
            self.cond.notify_all()
        self.barrier.abort()

    def _check(self, rank):
        if self.aborted:
            raise RuntimeError('rank %d: another rank failed, the exchange was abandoned' % rank)

    def allreduce(self, rank, values):
        """values: numpy array or torch tensor, summed over the ranks in place."""
        is_numpy = isinstance(values, np.ndarray)
        self.slots[rank] = values.copy() if is_numpy else values.clone()
        self.barrier.wait(self.timeout)
        total = self.slots[0].copy() if is_numpy else self.slots[0].clone()
        for r in range(1, self.world):
            total += self.slots[r]
        self.barrier.wait(self.timeout)             # every rank has read the slots before anyone overwrites one
        if is_numpy:
            values[:] = total
        else:
            values.copy_(total)
        if rank == 0:
            self.reduces += 1

    def exchange(self, rank, sends, recvs):
        """sends = [(peer, array)], recvs = [(peer, array to fill)]; numpy arrays are copied at send time, device tensors are handed
        over by reference and copied by the receiver."""
        import time
        deadline = time.time() + self.timeout
        with self.cond:
            for peer, h in sends:
                by_ref = not isinstance(h, np.ndarray)
                self.mail.setdefault((rank, peer), []).append(h if by_ref else h.copy())
                self.pending[rank] += 1 if by_ref else 0
                self.messages += 1
            self.cond.notify_all()
        for peer, h in recvs:
            with self.cond:
                while not self.mail.get((peer, rank)):
                    self._check(rank)
                    if not self.cond.wait(max(0.0, deadline - time.time())) and time.time() >= deadline:
                        raise TimeoutError('rank %d: nothing from rank %d' % (rank, peer))
                src = self.mail[(peer, rank)].pop(0)
            if not isinstance(src, np.ndarray):
                h.copy_(src)
                import torch
                torch.cuda.synchronize(h.device)
                with self.cond:
                    self.pending[peer] -= 1
                    self.cond.notify_all()
            else:
                h[:] = src
        with self.cond:                             # device mode: my pack buffers are free again only when my messages were copied
            while self.pending[rank] > 0:
                self._check(rank)
                if not self.cond.wait(max(0.0, deadline - time.time())) and time.time() >= deadline:
                    raise TimeoutError('rank %d: %d message(s) never consumed' % (rank, self.pending[rank]))


class LocalComm:
    """The Comm interface of the phase-by-phase driver (TiledTransfer: Adam or L-BFGS) over an InProcessFabric: ranks are threads of
    this process on one GPU, tensors are exchanged by reference and copied device to device."""

    def __init__(self, fabric, rank):
        self.fabric, self.rank, self.world = fabric, rank, fabric.world
        self.dist, self.staged = None, False

    def all_reduce(self, t):
        if self.world == 1 or t is None or t.numel() == 0:
            return t
        self.fabric.allreduce(self.rank, t)
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        return t

    def exchange(self, sends, recvs):
        if self.world > 1:
            self.fabric.exchange(self.rank, list(sends), list(recvs))


def run_in_process(ranks, steps, fabric, on_step=None):
    """`steps` iterations of every FusedTiledTransfer in `ranks`, one thread per rank.  Returns, per rank, the trace values of every step
    (on_step(rank, step, transfer, values) may collect more, e.g. the tile image).  A rank that raises aborts the fabric, so the others
    leave their waits at once; if a rank thread is nevertheless still running when this returns (it sits inside the engine), the
    RuntimeError carries ``still_running = True`` and the caller must NOT free the engine contexts."""
    import threading
    world = len(ranks)
    out, errors = [None] * world, []

    def run(r):
        try:
            res = []
            for k in range(steps):
                vals = ranks[r].step()
                res.append(on_step(r, k, ranks[r], vals) if on_step else vals)
            out[r] = res
        except Exception as e:          # noqa: BLE001
            errors.append((r, repr(e)))
            fabric.abort() if hasattr(fabric, 'abort') else fabric.barrier.abort()
    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(max(300.0, 4 * fabric.timeout))
    alive = [r for r, t in enumerate(threads) if t.is_alive()]
    if alive and hasattr(fabric, 'abort'):
        fabric.abort()
        for t in threads:
            t.join(30.0)
        alive = [r for r, t in enumerate(threads) if t.is_alive()]
    if errors or alive or any(o is None for o in out):
        err = RuntimeError('in-process ranks failed: %s' % (errors or ('rank(s) %s did not finish' % alive)))
        err.still_running = bool(alive)
        raise err
    return out


def rendezvous_unique_id(rank, world, make_id, addr=None, port=None, timeout=None):
    """Hands rank 0's 128-byte communicator id to every rank over a plain TCP socket (MASTER_ADDR : MASTER_PORT + 17 of the launcher's
    environment): the control plane of st_comm_init, so that the RCCL path needs neither torch.distributed nor a second HIP runtime.
    Every wait is bounded by `timeout` seconds (default: ST2_RENDEZVOUS_TIMEOUT_S, else 120) and ends in a TimeoutError that names the
    rank and what it was waiting for -- a rank that never arrives must not leave the others hanging in ncclCommInitRank."""
    import os
    import socket
    import time
    if world == 1:
        return make_id()
    if timeout is None:
        timeout = float(os.environ.get('ST2_RENDEZVOUS_TIMEOUT_S', '120'))
    addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(port if port is not None else int(os.environ.get('MASTER_PORT', '29500')) + 17)
    deadline = time.time() + timeout
    if rank == 0:
        uid = make_id()
        srv = socket.socket()
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        served = 0
        try:
            while served < world - 1:
                srv.settimeout(max(0.05, deadline - time.time()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError('rank 0: only %d of %d ranks fetched the communicator id from %s:%d within %g s'
                                       % (served, world - 1, addr, port, timeout)) from None
                conn.settimeout(10.0)
                conn.sendall(uid)
                conn.close()
                served += 1
        finally:
            srv.close()
        return uid
    while True:
        try:
            conn = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise TimeoutError('rank %d: rank 0 never opened the communicator-id rendezvous at %s:%d within %g s' % (rank, addr, port, timeout)) from None
            time.sleep(0.2)
    uid = b''
    try:
        while len(uid) < 128:
            conn.settimeout(max(0.05, deadline - time.time()))
            try:
                chunk = conn.recv(128 - len(uid))
            except socket.timeout:
                raise TimeoutError('rank %d: the communicator id did not arrive from %s:%d within %g s' % (rank, addr, port, timeout)) from None
            if not chunk:
                raise ConnectionError('rank 0 closed the rendezvous early')
            uid += chunk
    finally:
        conn.close()
    return uid


def crop(image, rect):
    """HxWx3 array -> the window/tile crop."""
    return np.ascontiguousarray(image[rect.y0:rect.y1, rect.x0:rect.x1])
