"""HIP tile backend of the tile-sharded mode: one ``Engine`` per rank holding that rank's window in HBM;
the phases of tiled.TiledTransfer map 1:1 onto the ``st_tile_*`` entry points of the C ABI.  Device buffers
are handed to torch (for RCCL) as zero-copy views through ``__cuda_array_interface__``."""

import ctypes
from ctypes import byref, c_int, c_void_p

import numpy as np

from . import capi
from .capi import check
from .engine import Engine, OPT_ADAM, OPT_LBFGS
from .tiling import blob_geometry
from .transfer import weight_table, LOSS_NAMES, SCALAR_LOSS_NAMES, EPS

F32 = np.float32


class _DevArray:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': '<f4', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def dev_tensor(ptr, shape, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, shape), device=device)


class _LazyTorch:
    """torch is needed only by the phase-by-phase driver (tiled.TiledTransfer: device buffers handed to torch.distributed) and by the
    host-staged test transport.  The in-engine iteration (st_tile_step over RCCL) never imports it: one HIP runtime in the process."""
    def __getattr__(self, name):
        import torch
        return getattr(torch, name)


torch = _LazyTorch()


class HipTileBackend:
    def __init__(self, net_params, grid, rank, content, style, init, weights, params, step_size=10,
                 topology=None, device=0, precision='fp32', use_torch=True, optimizer='adam'):
        # torch bundles its own copy of the HIP runtime; when both live in one process torch's must come up FIRST.
        # The phase-by-phase driver and the host-staged test transport need torch; the in-engine iteration over RCCL
        # (comm_init_rccl + tiled.FusedTiledTransfer) does not: pass use_torch=False and the process never loads it.
        if use_torch:
            import torch as _torch
            if _torch.cuda.is_available():
                _torch.cuda.init()
        # precision='bf16': the convs of the window run on the bf16 matrix cores (fp32 accumulation) with the lean data flow (fp32 blobs /
        # diffs only where something reads fp32); the Gram / style / loss kernels of the tile phases stay fp32 on the fp32 blobs of the
        # weighted layers (the region-of-interest forms exist for those only).  'bf16-full' writes every fp32 tensor: same results.
        self.engine = Engine(topology, device, precision)
        self.engine.load_weights(net_params)
        self.lib, self.ctx = self.engine.lib, self.engine._ctx
        self.device_index = device
        self.grid, self.rank = grid, rank
        w, t = grid.windows[rank], grid.tiles[rank]
        crop = lambda im, r: np.ascontiguousarray(im[r.y0:r.y1, r.x0:r.x1])
        self.engine.set_input(crop(init, w))
        self.engine.set_content(crop(content, w))
        self.engine.set_style(style)
        rows, cells = weight_table(weights)
        cols = [[cells[k][r] for r in rows] for k in LOSS_NAMES]
        self.engine.set_weights(rows, cols[0], cols[1], cols[2], [params[k] for k in SCALAR_LOSS_NAMES])
        # optimizer: what the fused iteration (st_tile_step) runs -- Adam, or L-BFGS in its Gram form with one all-reduce of the new inner
        # products per step; the phase-by-phase driver (tiled.TiledTransfer) brings its own L-BFGS and only uses Adam's fused update
        if optimizer not in ('adam', 'lbfgs'):
            raise ValueError('optimizer must be adam or lbfgs')
        self.engine.optimizer_reset(OPT_LBFGS if optimizer == 'lbfgs' else OPT_ADAM, step_size)
        self.engine.clear_norms()
        check(self.lib.st_tile_configure(self.ctx, grid.gH, grid.gW, w.y0, w.x0, t.y0, t.x0, t.y1, t.x1))
        self.params = params
        self.wh, self.ww = w.y1 - w.y0, w.x1 - w.x0
        topo = self.engine.topology
        names = self.engine.blob_names
        ggeo = blob_geometry(topo, grid.gH, grid.gW)
        self.active = []
        for r in rows:
            cw, sw, dw = (cells[k][r] for k in LOSS_NAMES)
            flags = tuple(abs(v) > EPS for v in (cw, sw, dw))
            if any(flags):
                c, gh, gw, _ = ggeo[names.index(r)]
                self.active.append((r, names.index(r), F32(cw), F32(sw), F32(dw)) + flags + (c, c * gh * gw))
        self.n_style = sum(1 for a in self.active if a[6])

    @property
    def device(self):
        return torch.device('cuda', self.device_index)

    def _sync(self):
        self.engine.sync()

    # ---- the iteration with its communication inside the engine (st_tile_step; engine_comm.cpp) -------------------------------
    def comm_init_rccl(self, unique_id, rank, world):
        """One RCCL communicator for this context (unique_id: the 128 bytes of st_comm_unique_id from rank 0)."""
        check(self.lib.st_comm_init(self.ctx, bytes(unique_id), int(rank), int(world)))

    def comm_init_host(self, rank, world, allreduce, exchange):
        """Host-staged transport in place of RCCL (tests: several ranks share one GPU).  `allreduce(values)` sums a float32 numpy
        array over the ranks in place; `exchange(sends, recvs)` delivers sends = [(peer, float32 array)] and fills
        recvs = [(peer, float32 array to fill)].  The engine calls them from st_tile_step with its stream drained."""
        device = self.device

        def on_allreduce(_user, ptr, n):
            try:
                t = dev_tensor(ptr, (n,), device)
                h = t.cpu().numpy()
                allreduce(h)
                t.copy_(torch.from_numpy(h))
                torch.cuda.synchronize(device)
                return 0
            except Exception:               # noqa: BLE001 (reported through the status code)
                import traceback
                traceback.print_exc()
                return 1

        def on_exchange(_user, ns, speer, sbuf, scount, nr, rpeer, rbuf, rcount):
            try:
                sends = [(int(speer[i]), dev_tensor(sbuf[i], (scount[i],), device).cpu().numpy()) for i in range(ns)]
                recvs = [(int(rpeer[j]), np.empty(rcount[j], F32)) for j in range(nr)]
                exchange(sends, recvs)
                for j, (_, h) in enumerate(recvs):
                    dev_tensor(rbuf[j], (rcount[j],), device).copy_(torch.from_numpy(h))
                torch.cuda.synchronize(device)
                return 0
            except Exception:               # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1
        self._callbacks = (capi.ALLREDUCE_FN(on_allreduce), capi.EXCHANGE_FN(on_exchange))       # keep the thunks alive
        check(self.lib.st_comm_callbacks(self.ctx, int(rank), int(world), self._callbacks[0], self._callbacks[1], None))

    def comm_init_local(self, rank, world, fabric):
        """Ranks that are threads of this process on this GPU (tiled.InProcessFabric): the engine's all-reduces and strip exchanges
        become device-to-device copies between the contexts' buffers -- no host staging."""
        device = self.device

        def on_allreduce(_user, ptr, n):
            try:
                fabric.allreduce(rank, dev_tensor(ptr, (n,), device))
                torch.cuda.synchronize(device)
                return 0
            except Exception:               # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        def on_exchange(_user, ns, speer, sbuf, scount, nr, rpeer, rbuf, rcount):
            try:
                sends = [(int(speer[i]), dev_tensor(sbuf[i], (scount[i],), device)) for i in range(ns)]
                recvs = [(int(rpeer[j]), dev_tensor(rbuf[j], (rcount[j],), device)) for j in range(nr)]
                fabric.exchange(rank, sends, recvs)
                return 0
            except Exception:               # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1
        self._callbacks = (capi.ALLREDUCE_FN(on_allreduce), capi.EXCHANGE_FN(on_exchange))       # keep the thunks alive
        check(self.lib.st_comm_callbacks(self.ctx, int(rank), int(world), self._callbacks[0], self._callbacks[1], None))

    def comm_init_callbacks(self, dist, rank, world):
        """comm_init_host over a torch.distributed (gloo) group: one process per rank."""
        def allreduce(values):
            t = torch.from_numpy(values)
            dist.all_reduce(t)

        def exchange(sends, recvs):
            ops = [dist.P2POp(dist.isend, torch.from_numpy(h), peer) for peer, h in sends]
            ops += [dist.P2POp(dist.irecv, torch.from_numpy(h), peer) for peer, h in recvs]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        self.comm_init_host(rank, world, allreduce, exchange)

    def comm_init_solo(self, rank, world):
        """ONE rank of a larger grid alone on a GPU (timing the per-rank compute of a multi-GPU run): the all-reduces see one rank,
        what the neighbours would send never arrives -- the receive buffers hold the zeros st_tile_plan initialised them with, so the
        timed update runs on finite data."""
        self._callbacks = (capi.ALLREDUCE_FN(lambda user, ptr, n: 0), capi.EXCHANGE_FN(lambda *a: 0))
        check(self.lib.st_comm_callbacks(self.ctx, int(rank), int(world), self._callbacks[0], self._callbacks[1], None))

    def set_plan(self, phase, peers):
        """peers: {peer rank: (send rects, recv rects)}, rects = [(y0, x0, h, w), ...] (tiled.fused_plans)."""
        keep = []
        arr = (capi.TilePeer * max(len(peers), 1))()
        for k, peer in enumerate(sorted(peers)):
            send, recv = peers[peer]
            sa = (c_int * max(4 * len(send), 1))(*[int(v) for r in send for v in r])
            ra = (c_int * max(4 * len(recv), 1))(*[int(v) for r in recv for v in r])
            keep += [sa, ra]
            arr[k].peer, arr[k].n_send, arr[k].n_recv = int(peer), len(send), len(recv)
            arr[k].send_rects = ctypes.cast(sa, ctypes.POINTER(c_int))
            arr[k].recv_rects = ctypes.cast(ra, ctypes.POINTER(c_int))
        check(self.lib.st_tile_plan(self.ctx, int(phase), len(peers), arr))

    def step_fused(self):
        """One iteration (Adam or L-BFGS, as the backend was built), every phase and collective enqueued by the engine; returns the
        trace values."""
        trace = np.zeros(self.engine.trace_len(), np.float64)
        check(self.lib.st_tile_step(self.ctx, trace.ctypes.data_as(c_void_p)))
        return trace

    def step_fused_async(self):
        """The same iteration without its trace: nothing is read back, so the call returns as soon as the launches and collectives
        are enqueued (over RCCL the host then runs ahead of the GPU, as Engine.step(want_trace=False) does on one GPU); the
        reduced sums stay on the device."""
        check(self.lib.st_tile_step(self.ctx, None))

    def barrier(self):
        check(self.lib.st_comm_barrier(self.ctx))

    def tile_image(self):
        t = self.grid.tiles[self.rank]
        out = np.empty((t.y1 - t.y0, t.x1 - t.x0, 3), F32)
        check(self.lib.st_tile_get_tile(self.ctx, out.ctypes.data_as(c_void_p)))
        return out

    def _buf(self, which, shape):
        p = c_void_p()
        check(self.lib.st_tile_buffer(self.ctx, which, byref(p)))
        return dev_tensor(p.value, shape, self.device)

    def x_cur(self):
        return self._buf(0, (3, self.wh, self.ww))

    def x_next(self):
        return self._buf(1, (3, self.wh, self.ww))

    def swap(self):
        torch.cuda.synchronize(self.device)
        check(self.lib.st_tile_swap(self.ctx))

    def strips(self, tensor, rects, buf, mode):
        """One kernel per neighbour and phase: pack (mode 0) the rects [(y0, x0, h, w)] of the (3, wh, ww) device tensor into
        the 1-D device buffer `buf`, or unpack them from it (1 assign, 2 add)."""
        torch.cuda.synchronize(self.device)
        flat = (c_int * (4 * len(rects)))(*[int(v) for r in rects for v in r])
        check(self.lib.st_tile_strips(self.ctx, c_void_p(tensor.data_ptr()), tensor.shape[0], tensor.shape[1], tensor.shape[2],
                                      len(rects), flat, c_void_p(buf.data_ptr()), mode))

    # ---- phases -------------------------------------------------------------------------------------------
    def forward_partials(self):
        torch.cuda.synchronize(self.device)
        p, n = c_void_p(), c_int()
        check(self.lib.st_tile_forward(self.ctx, byref(p), byref(n)))
        self._sync()
        self.p1 = dev_tensor(p.value, (n.value,), self.device) if n.value else None
        return self.p1

    def losses_need_style_norm(self):
        torch.cuda.synchronize(self.device)
        self.p1_host = self.p1.cpu().numpy().copy() if self.p1 is not None else np.zeros(0, F32)
        p, n = c_void_p(), c_int()
        check(self.lib.st_tile_losses(self.ctx, byref(p), byref(n)))
        self.p2 = None
        if n.value:
            check(self.lib.st_tile_style_raw(self.ctx))
            self._sync()
            self.p2 = dev_tensor(p.value, (n.value,), self.device)
        return self.p2

    def finish_losses(self):
        torch.cuda.synchronize(self.device)
        self.s2_host = self.p2.cpu().numpy().copy() if self.p2 is not None else None
        check(self.lib.st_tile_losses_finish(self.ctx))
        self._sync()

    def backward(self):
        p = c_void_p()
        check(self.lib.st_tile_backward(self.ctx, byref(p)))
        return dev_tensor(p.value, (3, self.wh, self.ww), self.device)

    def update(self, ring):
        ring = ring.contiguous()
        torch.cuda.synchronize(self.device)
        self._ring = ring                                    # keep alive until the kernel has run
        p, n = c_void_p(), c_int()
        check(self.lib.st_tile_update(self.ctx, c_void_p(ring.data_ptr()), byref(p), byref(n)))
        self.p3 = dev_tensor(p.value, (n.value,), self.device)
        return self.p3

    # ---- L-BFGS pieces (tiled.TiledTransfer, optimizer='lbfgs'): vectors are compact (3, th, tw) tile tensors ----------------
    def _tile_rect(self):
        t, w = self.grid.tiles[self.rank], self.grid.windows[self.rank]
        return [(t.y0 - w.y0, t.x0 - w.x0, t.y1 - t.y0, t.x1 - t.x0)]

    def gradient(self, ring):
        """The image-space pass without an optimizer (st_tile_gradient): partial sums as update() returns them; the combined
        gradient of the tile is then available from grad_tile()."""
        ring = ring.contiguous()
        torch.cuda.synchronize(self.device)
        self._ring = ring
        p, n = c_void_p(), c_int()
        check(self.lib.st_tile_gradient(self.ctx, c_void_p(ring.data_ptr()), byref(p), byref(n)))
        self.p3 = dev_tensor(p.value, (n.value,), self.device)
        return self.p3

    def vnew(self):
        t = self.grid.tiles[self.rank]
        return torch.empty((3, t.y1 - t.y0, t.x1 - t.x0), dtype=torch.float32, device=self.device)

    def grad_tile(self):
        out = self.vnew()
        self.strips(self._buf(4, (3, self.wh, self.ww)), self._tile_rect(), out.reshape(-1), 0)
        return out

    def vcopy(self, v):
        out = self.vnew()
        out.copy_(v)                                    # (device-to-device copy: memory plumbing)
        torch.cuda.synchronize(self.device)
        return out

    def vdot(self, a, b):
        """This rank's partial sum of <a, b> as a 1-element device tensor (utils.dot, utils.py:29-36)."""
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        check(self.lib.st_vec_dot(self.ctx, c_void_p(a.data_ptr()), c_void_p(b.data_ptr()), a.numel(), c_void_p(out.data_ptr())))
        return out

    def vaxpy(self, alpha, x, y):
        """y += alpha x (utils.axpy, utils.py:38-47)."""
        torch.cuda.synchronize(self.device)
        check(self.lib.st_vec_axpy(self.ctx, float(F32(alpha)), c_void_p(x.data_ptr()), c_void_p(y.data_ptr()), x.numel()))

    def vscale(self, alpha, y):
        """y *= alpha, as y = (alpha - 1) y + y would not round the same way: axpy onto a zeroed vector."""
        src = self.vcopy(y)
        y.zero_()
        self.vaxpy(alpha, src, y)

    def vdiv(self, divisor, y):
        """y /= divisor with the quotient formed in double and rounded once (optimizers.py:99: p /= np.sqrt(...), a float64)."""
        torch.cuda.synchronize(self.device)
        check(self.lib.st_vec_div(self.ctx, float(divisor), c_void_p(y.data_ptr()), y.numel()))

    def apply_step(self, s):
        """x_next[tile] = x_cur[tile] + s; the rest of x_next starts as a copy of x_cur (its apron is refreshed by the caller)."""
        xt = self.vnew()
        rect = self._tile_rect()
        self.strips(self.x_cur(), rect, xt.reshape(-1), 0)
        self.vaxpy(1.0, s, xt)
        nxt = self.x_next()
        nxt.copy_(self.x_cur())
        torch.cuda.synchronize(self.device)
        self.strips(nxt, rect, xt.reshape(-1), 1)

    def finish_trace(self):
        """Trace scalars from the reduced sums, in the reference's fp32 order (worker.py:249-301); same layout
        as the single-GPU engine: 6 per active layer + 8."""
        torch.cuda.synchronize(self.device)
        p3 = self.p3.cpu().numpy()
        s2 = self.s2_host if self.s2_host is not None else p3[6:]
        d2 = self._buf(2, (max(self.n_style, 1),)).cpu().numpy()
        norms = self._buf(3, (len(self.engine.blob_names) * 3,)).cpu().numpy()
        vals, loss, pos, k = [], F32(0), 0, 0
        for name, b, cw, sw, dw, c_on, s_on, d_on, C, n in self.active:
            sums = self.p1_host[pos:pos + 4]
            pos += 4
            v6 = [0.0] * 6
            if c_on:
                cn = norms[b * 3 + 0]
                v6[0] = cw * F32(sums[0] / n) / cn
                v6[1] = abs(cw) * np.sqrt(F32(sums[1] / n)) / cn
                loss += v6[0]
            if s_on:
                sn = norms[b * 3 + 1]
                v6[2] = sw * F32(d2[k] / (C * C)) / sn
                v6[3] = abs(sw / sn) * np.sqrt(F32(s2[k] / n))
                loss += v6[2]
                pos += C * C
                k += 1
            if d_on:
                dn = norms[b * 3 + 2]
                v6[4] = -dw * F32(sums[2] / n) / dn
                v6[5] = abs(dw) * np.sqrt(F32(sums[3] / n)) / dn
                loss += v6[4]
            vals += v6
        n3 = 3.0 * self.grid.gH * self.grid.gW
        scd_loss = loss
        t_loss = F32(self.params['tv']) * p3[0]
        p_loss = F32(self.params['p']) * (p3[1] / F32(self.params['p_power']))
        total = scd_loss + t_loss + p_loss
        vals += [scd_loss, t_loss, p_loss, np.sqrt(F32(p3[2] / n3)), np.sqrt(F32(p3[3] / n3)),
                 np.sqrt(F32(p3[4] / n3)), total, np.sqrt(F32(p3[5] / n3))]
        return np.asarray(vals, np.float64)
