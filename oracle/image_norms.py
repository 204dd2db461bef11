"""Image-space regularisers of the objective, restated on the CPU.  TEST INFRASTRUCTURE.

Follows utils.py:285-297 (``tv_norm``; circular shifts are utils.py:232-254 ``roll_by_one``)
and utils.py:300-304 (``p_norm``).  Callers pass u = x / 255 (worker.py:283,287) and multiply
the returned gradient by the weight WITHOUT a 1/255 chain factor (worker.py:296-297).
Pinned against the reference's own functions by tests/golden/image_norms.npz.
"""

import numpy as np


def tv_term(u, beta=2):
    """Total-variation value and gradient of a (1, C, H, W) float32 array, periodic borders.

    a = u - u[col+1], b = u - u[row+1] (wrapping); q = a^2 + b^2 + 1e-8;
    value = sum q^(beta/2); gradient = da + db - da[col-1] - db[row-1] with
    da = 2 a k, db = 2 b k, k = (beta/2) q^(beta/2 - 1).
    """
    a = u - np.roll(u, -1, axis=3)
    b = u - np.roll(u, -1, axis=2)
    q = a**2 + b**2 + 1e-8
    value = np.sum(q**(beta / 2))
    k = (beta / 2) * q**(beta / 2 - 1)
    da = 2 * a * k
    db = 2 * b * k
    g = da + db
    g -= np.roll(da, 1, axis=3)
    g -= np.roll(db, 1, axis=2)
    return value, g


def p_term(u, p=2):
    """sum |u|^p / p and its gradient sign(u) |u|^(p-1)."""
    mag = abs(u)
    return np.sum(mag**p) / p, np.sign(u) * mag**(p - 1)
