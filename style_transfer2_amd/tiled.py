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
import torch

from .tiling import Rect


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
    def __init__(self, grid, rank, backend, comm):
        self.grid, self.rank, self.backend, self.comm = grid, rank, backend, comm
        self.window, self.tile = grid.windows[rank], grid.tiles[rank]
        self.t = 0
        self._refresh = grid.apron_refresh_plan()
        self._overlap = grid.grad_overlap_plan()
        self._ring = grid.ring_plan()

    # -- communication phases ------------------------------------------------------------------------------
    def refresh_aprons(self, x):
        """x: (3, wh, ww) window tensor whose TILE part is current: fill the apron from the owners."""
        sends, recvs, dst_slices = [], [], []
        for src, dst, rect in self._refresh:
            if src == self.rank:
                ys, xs = _local(rect, self.window)
                sends.append((dst, x[:, ys, xs].contiguous()))
            elif dst == self.rank:
                ys, xs = _local(rect, self.window)
                buf = torch.empty((3, rect.y1 - rect.y0, rect.x1 - rect.x0), dtype=x.dtype, device=x.device)
                recvs.append((src, buf))
                dst_slices.append((ys, xs, buf))
        self.comm.exchange(sends, recvs)
        for ys, xs, buf in dst_slices:
            x[:, ys, xs] = buf

    def overlap_add(self, g):
        """g: (3, wh, ww) window gradient: add the neighbours' contributions to MY tile pixels."""
        sends, recvs, adds = [], [], []
        for src, dst, rect in self._overlap:
            if src == self.rank:
                ys, xs = _local(rect, self.window)
                sends.append((dst, g[:, ys, xs].contiguous()))
            elif dst == self.rank:
                ys, xs = _local(rect, self.window)
                buf = torch.empty((3, rect.y1 - rect.y0, rect.x1 - rect.x0), dtype=g.dtype, device=g.device)
                recvs.append((src, buf))
                adds.append((ys, xs, buf))
        self.comm.exchange(sends, recvs)
        for ys, xs, buf in adds:                 # fixed plan order -> deterministic sums
            g[:, ys, xs] += buf

    def gather_ring(self, x):
        """(3, th+2, tw+2) tensor: the tile's 1-px neighbourhood under the image's periodic wrap."""
        th, tw = self.tile.y1 - self.tile.y0, self.tile.x1 - self.tile.x0
        ring = torch.zeros((3, th + 2, tw + 2), dtype=x.dtype, device=x.device)
        sends, recvs, places = [], [], []
        for dst, items in enumerate(self._ring):
            for src, rect, ry, rx in items:
                h, w = rect.y1 - rect.y0, rect.x1 - rect.x0
                if src == self.rank:
                    ys, xs = _local(rect, self.window)      # my own tile pixels, in my window
                    piece = x[:, ys, xs]
                    if dst == self.rank:
                        ring[:, ry:ry + h, rx:rx + w] = piece
                    else:
                        sends.append((dst, piece.contiguous()))
                elif dst == self.rank:
                    buf = torch.empty((3, h, w), dtype=x.dtype, device=x.device)
                    recvs.append((src, buf))
                    places.append((ry, rx, h, w, buf))
        self.comm.exchange(sends, recvs)
        for ry, rx, h, w, buf in places:
            ring[:, ry:ry + h, rx:rx + w] = buf
        return ring

    # -- one iteration ----------------------------------------------------------------------------------------
    def step(self):
        b = self.backend
        self.t += 1
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


def crop(image, rect):
    """HxWx3 array -> the window/tile crop."""
    return np.ascontiguousarray(image[rect.y0:rect.y1, rect.x0:rect.x1])
