"""Host-side Pillow resampling of fp32 planes (reference utils.py:130-160).

Not on the per-iteration path: used only by SetImages.RESAMPLE and by
a change of input size with a live optimizer.  Pillow, exactly as the reference uses it."""

from concurrent.futures import ThreadPoolExecutor
import os

import numpy as np
from PIL import Image

F32 = np.float32
LANCZOS = Image.LANCZOS
BILINEAR = Image.BILINEAR


def _plane(src, dst, hw, method):
    dst[:] = Image.fromarray(src).resize((hw[1], hw[0]), method)


def resample_nchw(a, hw, method=LANCZOS):
    a = np.asarray(a, F32)
    n, ch = a.shape[:2]
    out = np.zeros((n, ch, hw[0], hw[1]), F32)
    with ThreadPoolExecutor(max_workers=os.cpu_count()) as pool:
        futures = [pool.submit(_plane, a[i, j], out[i, j], hw, method) for i in range(n) for j in range(ch)]
        for fut in futures:
            fut.result()
    return out


# ---------------------------------------------------------------------------------------------------------
# Pillow's coefficient tables, restated (Pillow Resample.c precompute_coeffs; Pillow is the third-party library
# the reference calls, utils.py:131).  The device kernels consume these tables, so the GPU path is bit-exact with
# ``Image.resize`` on mode-'F' planes (checked in tests/test_resample.py against Pillow itself).
# ---------------------------------------------------------------------------------------------------------
import math


def _lanczos3(x):
    x = abs(x)
    if x >= 3.0:
        return 0.0
    if x == 0.0:
        return 1.0
    xp = x * math.pi
    return 3.0 * math.sin(xp) * math.sin(xp / 3.0) / (xp * xp)


def _triangle(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


_FILTERS = {LANCZOS: (_lanczos3, 3.0), BILINEAR: (_triangle, 1.0)}


def pillow_coeffs(in_size, out_size, method=LANCZOS):
    """(lo int32[out], n int32[out], k float64[out, kmax]) for one axis."""
    filt, support = _FILTERS[method]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support * filterscale
    kmax = int(math.ceil(support)) * 2 + 1
    lo = np.zeros(out_size, np.int32)
    n = np.zeros(out_size, np.int32)
    k = np.zeros((out_size, kmax), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        total = 0.0
        for v in w:
            total += v
        if total != 0.0:
            w = [v / total for v in w]
        lo[xx], n[xx] = xmin, xmax
        k[xx, :xmax] = w
    return lo, n, k


def resample_planes_reference(a, hw, method=LANCZOS):
    """Numpy restatement of Pillow's two-pass float resize (horizontal, then vertical; double accumulation in
    window order; float32 intermediate).  Used to pin ``pillow_coeffs`` on the CPU."""
    a = np.asarray(a, F32)
    h, w = a.shape[-2:]
    flat = a.reshape(-1, h, w)
    cur = flat
    if w != hw[1]:
        lo, n, k = pillow_coeffs(w, hw[1], method)
        out = np.zeros((flat.shape[0], h, hw[1]), F32)
        for xx in range(hw[1]):
            acc = np.zeros((flat.shape[0], h), np.float64)
            for i in range(n[xx]):
                acc += cur[:, :, lo[xx] + i].astype(np.float64) * k[xx, i]
            out[:, :, xx] = acc
        cur = out
    if h != hw[0]:
        lo, n, k = pillow_coeffs(h, hw[0], method)
        out = np.zeros((flat.shape[0], hw[0], hw[1]), F32)
        for yy in range(hw[0]):
            acc = np.zeros((flat.shape[0], hw[1]), np.float64)
            for i in range(n[yy]):
                acc += cur[:, lo[yy] + i, :].astype(np.float64) * k[yy, i]
            out[:, yy, :] = acc
        cur = out
    return cur.reshape(a.shape[:-2] + tuple(hw)).copy()
