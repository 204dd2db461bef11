"""Image-space receptive fields of blob positions.  TEST INFRASTRUCTURE (tests/ and bench.py's parity leg).

ReLU and max-pool are discontinuous: two correct fp32 forwards take different branches at a few activations per hundred
million, and each such flip changes the gradient by O(1) -- but only inside the image-space receptive field of the flipped
unit (models/vgg19.prototxt: 3x3 / pad 1 convolutions, 2x2 / stride 2 pools).  The parity checks paint those fields and hold
everything outside them to the gate."""
import numpy as np  # noqa: F401


def receptive_geometry(topology):
    """Per blob (0 = data): (stride, lo, hi) such that unit y of the blob sees image rows [stride * y - lo, stride * y + hi]
    (3x3 / pad 1 convolutions, 2x2 / stride 2 pools; same along x)."""
    geo = [(1, 0, 0)]
    a, lo, hi = 1, 0, 0
    for layer in topology:
        if layer[0] == 'conv':
            lo, hi = lo + a, hi + a
        else:
            hi, a = hi + a, a * 2
        geo.append((a, lo, hi))
    return geo


def paint_receptive_fields(mask, positions, geom):
    """mask (H, W) bool: set the image-space receptive field of every blob position (y, x) in `positions` ((n, 2) ints)."""
    a, lo, hi = geom
    h, w = mask.shape
    for y, x in positions:
        mask[max(0, a * y - lo):min(h, a * y + hi + 1), max(0, a * x - lo):min(w, a * x + hi + 1)] = True
    return mask
