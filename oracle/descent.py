"""The two image optimisers of the reference, restated on the CPU.  TEST INFRASTRUCTURE.

Follows optimizers.py:7-46 (Adam), optimizers.py:49-125 (fixed-step L-BFGS),
utils.py:49-69 (``DecayingMean``) and utils.py:29-46 (BLAS-1 ``dot`` / ``axpy``).
Pinned against the reference's own classes by tests/golden/descent.npz and
tests/golden/transfer_*.npz.
"""

import numpy as np
from scipy.linalg import blas


def sdot(x, y):
    """fp32 BLAS dot of two equally shaped arrays (utils.py:29-34)."""
    if x.shape != y.shape:
        raise ValueError('Sizes do not match: x=%s y=%s' % (x.shape, y.shape))
    return blas.sdot(x.ravel(), y.ravel())


def saxpy(a, x, y):
    """y <- a x + y in place, fp32 BLAS (utils.py:38-46)."""
    if x.shape != y.shape:
        raise ValueError('Sizes do not match: x=%s y=%s' % (x.shape, y.shape))
    out = blas.saxpy(x.ravel(), y.ravel(), a=a).reshape(y.shape)
    if out is not y:
        y[:] = out
    return y


class EmaBiasCorrected:
    """Exponential moving average with start-up bias correction (utils.py:49-69)."""

    def __init__(self, decay):
        self.decay = decay
        self.mean = 0
        self.items = 0

    def push(self, item):
        self.mean = self.decay * self.mean + (1 - self.decay) * item
        self.items += 1

    def value(self):
        if self.items == 0:
            return self.mean
        return self.mean / (1 - self.decay**self.items)

    def clear(self):
        self.mean = 0
        self.items = 0


class AdamOracle:
    """optimizers.py:7-46.  x is updated in place; ``objective_changed`` clears m only."""

    def __init__(self, x, opfunc, step_size=1, b1=0.9, b2=0.999):
        self.x, self.opfunc, self.step_size = x, opfunc, step_size
        self.t = 0
        self.g1 = EmaBiasCorrected(b1)
        self.g2 = EmaBiasCorrected(b2)

    def step(self):
        self.t += 1
        loss, grad = self.opfunc(self.x)
        self.g1.push(grad)
        self.g2.push(grad**2)
        self.x -= self.step_size * self.g1.value() / (np.sqrt(self.g2.value()) + 1e-8)
        return self.x, loss

    def objective_changed(self):
        self.t = 0
        self.g1.clear()


class LBFGSOracle:
    """optimizers.py:49-125: two-loop recursion, <= n_corr pairs, fixed step, no line search."""

    def __init__(self, x, opfunc, step_size=1, n_corr=10):
        self.x, self.opfunc, self.step_size, self.n_corr = x, opfunc, step_size, n_corr
        self.loss = self.grad = None
        self.pairs = []          # (s, y, s.y), oldest first

    def step(self):
        if self.loss is None:
            self.loss, self.grad = self.opfunc(self.x)
        s = -self.step_size * self.inv_hessian_times(self.grad)
        self.x += s
        loss, grad = self.opfunc(self.x)
        y = grad - self.grad
        sy = sdot(s, y)
        if sy > 1e-10:
            self.pairs.append((s, y, sy))
        if len(self.pairs) > self.n_corr:
            self.pairs = self.pairs[1:]
        self.loss, self.grad = loss, grad
        return self.x, loss

    def inv_hessian_times(self, p):
        p = p.copy()
        alphas = []
        for s, y, sy in reversed(self.pairs):
            alphas.append(sdot(s, p) / sy)
            saxpy(-alphas[-1], y, p)
        if self.pairs:
            _, y, sy = self.pairs[-1]
            p *= sy / sdot(y, y)
        else:
            p /= np.sqrt(sdot(p, p) / p.size)
        for (s, y, sy), alpha in zip(self.pairs, reversed(alphas)):
            beta = sdot(y, p) / sy
            saxpy(alpha - beta, s, p)
        return p

    def objective_changed(self):
        self.pairs = []
        self.loss = self.grad = None
