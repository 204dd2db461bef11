"""Tile-sharding of ONE large image over several GPUs (BASELINE config 5).

Design ("apron", no per-layer exchange): the image is cut into a grid of tiles whose edges are multiples of
the network's total stride (16 for VGG19 up to conv5_1).  Each rank runs the UNCHANGED network kernels on its
tile plus an apron wide enough to contain the receptive field of the deepest weighted layer (70 px -> 80),
clipped at the true image border (where the reference's zero padding applies anyway).  Every activation on a
path to a feature inside the tile is then bit-for-bit what a single-GPU run would compute, so only
reductions and image-space borders need communication:

  forward   : per-rank partial sums over the tile region (ROI) of each weighted blob
              -> ONE all-reduce of [scalars | raw Gram sums]          (Gram normalised by the GLOBAL n, worker.py:114)
  backward  : local, injected diffs are zero outside the ROI; gradient that lands on apron pixels belongs
              to the neighbour -> overlap-add exchange of window-gradient strips
  update    : TV uses the image's PERIODIC wrap (utils.py:232-254) -> a 1-px ring gathered from the torus
              neighbours; Adam is local to the tile; the refreshed tile pixels are sent to the windows that
              contain them (apron refresh)
  trace     : one more all-reduce of the image-space partial sums

This module is pure geometry + communication planning (no device code) and is shared by the HIP tile
backend and by the CPU test backend, so the distributed algorithm is covered by gloo tests without a GPU.
"""

from collections import namedtuple

Rect = namedtuple('Rect', 'y0 x0 y1 x1')          # half-open, global pixel coordinates


def rect_and(a, b):
    r = Rect(max(a.y0, b.y0), max(a.x0, b.x0), min(a.y1, b.y1), min(a.x1, b.x1))
    return r if r.y1 > r.y0 and r.x1 > r.x0 else None


def total_stride(topology, last_blob):
    s = 1
    for layer in topology[:last_blob]:
        if layer[0] == 'pool':
            s *= 2
    return s


def receptive_apron(topology, last_blob):
    """Input pixels a feature of blob `last_blob` reaches beyond its own stride-cell, rounded up to a multiple
    of the total stride (so that window origins keep every pooling window aligned with the global one)."""
    lo = hi = 0
    s = 1
    for layer in topology[:last_blob]:
        if layer[0] == 'conv':
            lo += s
            hi += s
        else:
            hi += s
            s *= 2
    need = max(lo, hi - (s - 1))
    return (need + s - 1) // s * s if need else 0


def split_edges(n, parts, quantum):
    """`parts` contiguous segments of [0, n) with inner edges on multiples of `quantum`."""
    edges = [0]
    for i in range(1, parts):
        e = int(round(n * i / parts / quantum)) * quantum
        edges.append(min(max(e, edges[-1] + quantum), n))
    edges.append(n)
    if any(b <= a for a, b in zip(edges, edges[1:])):
        raise ValueError('image of %d px cannot be cut into %d tiles of >= %d px' % (n, parts, quantum))
    return edges


class TileGrid:
    """Geometry of an R x C tiling of a gH x gW image for a given topology / deepest weighted blob."""

    def __init__(self, gH, gW, rows, cols, topology, last_blob, apron=None):
        self.gH, self.gW, self.rows, self.cols = gH, gW, rows, cols
        self.stride = total_stride(topology, last_blob)
        self.apron = receptive_apron(topology, last_blob) if apron is None else apron
        if self.apron % self.stride:
            raise ValueError('apron must be a multiple of the total stride')
        ey = split_edges(gH, rows, self.stride)
        ex = split_edges(gW, cols, self.stride)
        self.tiles, self.windows = [], []
        for r in range(rows):
            for c in range(cols):
                t = Rect(ey[r], ex[c], ey[r + 1], ex[c + 1])
                self.tiles.append(t)
                self.windows.append(Rect(max(0, t.y0 - self.apron), max(0, t.x0 - self.apron),
                                         min(gH, t.y1 + self.apron), min(gW, t.x1 + self.apron)))
        self.world = rows * cols

    def owner_pieces(self, rect):
        """Decompose a global rect into (owner rank, sub-rect) pieces along tile boundaries."""
        out = []
        for rank, t in enumerate(self.tiles):
            p = rect_and(rect, t)
            if p:
                out.append((rank, p))
        return out

    # ---- the three exchanges, as lists of transfers (src rank, dst rank, global rect, tag) -------------
    def apron_refresh_plan(self):
        """x of tile pixels -> every other rank's window that contains them."""
        plan = []
        for dst, w in enumerate(self.windows):
            for src, t in enumerate(self.tiles):
                if src != dst:
                    p = rect_and(w, t)
                    if p:
                        plan.append((src, dst, p))
        return plan

    def grad_overlap_plan(self):
        """window gradient of rank src over (window_src AND tile_dst) -> added into dst's gradient."""
        plan = []
        for src, w in enumerate(self.windows):
            for dst, t in enumerate(self.tiles):
                if src != dst:
                    p = rect_and(w, t)
                    if p:
                        plan.append((src, dst, p))
        return plan

    def ring_plan(self):
        """The 1-px ring around each tile under PERIODIC wrap of the whole image.  Returns, per destination
        rank, a list of (src rank, src global rect, ring y offset, ring x offset): ring arrays have shape
        (th + 2, tw + 2) with the tile at [1:-1, 1:-1]; only the border cells are ever read."""
        plans = []
        for dst, t in enumerate(self.tiles):
            th, tw = t.y1 - t.y0, t.x1 - t.x0
            items = []
            segs = [(t.y0 - 1, t.x0 - 1, 1, tw + 2, 0, 0), (t.y1, t.x0 - 1, 1, tw + 2, th + 1, 0),   # top, bottom rows
                    (t.y0, t.x0 - 1, th, 1, 1, 0), (t.y0, t.x1, th, 1, 1, tw + 1)]                   # left, right cols
            for gy, gx, h, w, ry, rx in segs:
                # wrap, then split into pieces that do not cross the image edge
                for (py, ph, oy) in _wrap_1d(gy, h, self.gH):
                    for (px, pw, ox) in _wrap_1d(gx, w, self.gW):
                        for src, piece in self.owner_pieces(Rect(py, px, py + ph, px + pw)):
                            items.append((src, piece, ry + oy + (piece.y0 - py), rx + ox + (piece.x0 - px)))
            plans.append(items)
        return plans

    def roi_in_blob(self, rank, blob_hw, blob_stride):
        """Tile region of `rank` in the LOCAL coordinates of a blob of its window with the given stride.
        blob_hw: (h, w) of the window's blob (to clamp the ragged last row/col at the image border)."""
        t, w = self.tiles[rank], self.windows[rank]
        s = blob_stride
        y0, x0 = (t.y0 - w.y0) // s, (t.x0 - w.x0) // s
        y1 = blob_hw[0] if t.y1 == self.gH else (t.y1 - w.y0) // s
        x1 = blob_hw[1] if t.x1 == self.gW else (t.x1 - w.x0) // s
        return Rect(y0, x0, y1, x1)


def _wrap_1d(start, length, n):
    """[(wrapped start, length, offset inside the original segment)] pieces of a periodic 1-D segment."""
    out, off = [], 0
    while length > 0:
        s = start % n
        take = min(length, n - s)
        out.append((s, take, off))
        start += take
        length -= take
        off += take
    return out


def pooled(n):
    q = (n - 2 + 1) // 2 if n >= 2 else 0
    return q + 1


def blob_geometry(topology, H, W):
    """[(C, h, w, stride)] for blob 0 .. len(topology) of an H x W input."""
    out = [(3, H, W, 1)]
    c, h, w, s = 3, H, W, 1
    for layer in topology:
        if layer[0] == 'conv':
            c = layer[3]
        else:
            h, w, s = pooled(h), pooled(w), s * 2
        out.append((c, h, w, s))
    return out
