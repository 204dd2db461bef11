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
