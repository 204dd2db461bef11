"""Wire messages between the web app and the style-transfer worker.

Pickle-compatible with the reference's ``messages.py`` (messages.py:38-172): pyzmq's
``send_pyobj``/``recv_pyobj`` pickle instances BY REFERENCE as ``messages.<ClassName>`` plus their
instance ``__dict__``, so what must match is the module name, the class names and the attribute
names/types -- they do (pinned by tests/golden/message_pickles.json, produced by the reference's own
classes).  ``app.py`` / ``router.py`` of the reference run against this module unchanged.
"""

import sys
import logging

import numpy as np

logger = logging.getLogger(__name__)


def _short(value):
    if isinstance(value, np.ndarray):
        return '<ndarray, shape: %s, dtype: %s>' % (value.shape, value.dtype)
    return repr(value)


class Message:
    """Base of every message: attribute bag + optional creation-site logging (``-dd``)."""
    debug = False

    def _set(self, **attrs):
        self.__dict__.update(attrs)
        if self.debug:
            # who constructed the message: two frames up (the constructor called _set); same log line as the reference's -dd output
            site = sys._getframe(2)
            logger.debug('%s created on line %d of %s: %s', type(self).__name__, site.f_lineno, site.f_code.co_filename, self)

    def __repr__(self):
        inner = ', '.join('%s=%s' % (k, _short(v)) for k, v in sorted(vars(self).items()))
        return '%s(%s)' % (type(self).__name__, inner)


# --- app <-> router (not used by the worker; kept so the module is a drop-in for both) ----------
class AppDown(Message):
    def __init__(self, addr, app_id):
        self._set(addr=addr, app_id=app_id)


class AppUp(Message):
    def __init__(self, addr, host, port, app_id):
        self._set(addr=addr, host=host, port=port, app_id=app_id)


class Reset(Message):
    def __init__(self):
        self._set()


# --- worker -> app ---------------------------------------------------------------------------------
class WorkerReady(Message):
    """First message of a worker: the model's layer (blob) names."""
    def __init__(self, layers=None):
        self._set(layers=[] if layers is None else layers)


class Iterate(Message):
    """One new iterate: HxWx3 float32 RGB image, iterate count, trace dict."""
    def __init__(self, image, i, trace):
        self._set(image=image, i=i, trace=trace)


class GetImages(Message):
    """The worker cannot iterate: some image slot is missing or sizes disagree; resend everything."""
    def __init__(self):
        self._set()


# --- app -> worker ---------------------------------------------------------------------------------
class SetImages(Message):
    """Fill image slots (HxWx3 RGB arrays).  ``None`` leaves a slot alone, ``RESAMPLE`` resamples the
    slot to ``size``; ``reset_state`` restarts the iterate count and the optimizer."""
    RESAMPLE = 1

    def __init__(self, size=None, input_image=None, content_image=None, style_image=None,
                 reset_state=False):
        self._set(size=size, input_image=input_image, content_image=content_image,
                  style_image=style_image, reset_state=reset_state)


def _optimizer_classes():
    try:
        from style_transfer2_amd.device_optimizers import AdamOptimizer, LBFGSOptimizer
        return {'adam': AdamOptimizer, 'lbfgs': LBFGSOptimizer}
    except ImportError:                      # the app side needs only the names
        return {'adam': 'adam', 'lbfgs': 'lbfgs'}


class SetOptimizer(Message):
    classes = _optimizer_classes()
    step_sizes = {'adam': 10, 'lbfgs': 1}

    def __init__(self, optimizer, step_size=None):
        if optimizer not in self.classes:
            raise ValueError('Invalid optimizer type')
        self._set(optimizer=optimizer, step_size=step_size if step_size else self.step_sizes[optimizer])


class SetWeights(Message):
    """``weights[loss][layer]`` for loss in ``loss_names``; ``params`` for the image-space terms."""
    loss_names = ('content', 'style', 'deepdream')
    scalar_loss_names = ('tv', 'tv_power', 'p', 'p_power')

    def __init__(self, weights, params):
        self._set(weights=weights, params=params)


class StartIteration(Message):
    def __init__(self):
        self._set()


class PauseIteration(Message):
    def __init__(self):
        self._set()


class Shutdown(Message):
    def __init__(self):
        self._set()
