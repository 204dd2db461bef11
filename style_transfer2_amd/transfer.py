"""``StyleTransfer``: host-side state machine of a style-transfer job, mirroring the reference class
of the same name (worker.py:117-315) method for method, with every tensor resident on the GPU.

What differs from the reference, by design:
  * ``input`` / ``features`` / ``grams`` live in HBM inside the engine; ``input`` is a property
    that downloads a copy, and so is ``content``; resampling (SetImages.RESAMPLE) runs on the device too;
  * ``opfunc`` and ``step`` run as device launches (forward, fused loss passes, ranged backward,
    fused TV/p-norm/Adam pass); the per-iteration trace is reduced on the device and read back
    with the iterate;
  * the loss-weight table is an ordered dict of dicts with ``DataFrame.from_dict`` row order
    instead of a pandas DataFrame (pinned by tests/golden/weight_order.json).
"""

from collections import OrderedDict
import csv
import time

import numpy as np

from .device_optimizers import AdamOptimizer, LBFGSOptimizer

F32 = np.float32
LOSS_NAMES = ('content', 'style', 'deepdream')            # reference messages.py:146
SCALAR_LOSS_NAMES = ('tv', 'tv_power', 'p', 'p_power')    # reference messages.py:147
EPS = 1e-15


def weight_table(weights):
    """Row order and cells of ``pd.DataFrame.from_dict(weights, dtype=np.float32)`` (reference
    worker.py:226-229): rows = union of the inner keys in order of first appearance, missing
    cells = NaN (which the |w| > 1e-15 tests treat as zero, worker.py:234)."""
    rows = []
    for kind in weights:
        for layer in weights[kind]:
            if layer not in rows:
                rows.append(layer)
    cells = {kind: OrderedDict((layer, F32(weights[kind].get(layer, np.nan))) for layer in rows)
             for kind in weights}
    return rows, cells


class Trace:
    """reference utils.py:257-282 (name de-duplication with '_', python scalars)"""

    def __init__(self):
        self.data = OrderedDict()

    def __call__(self, name, value):
        while name in self.data:
            name += '_'
        self.data[name] = float(value) if isinstance(value, (np.floating, float)) else int(value)
        return value

    def __str__(self):
        return ', '.join('%s: %g' % item for item in self.data.items())


class StyleTransfer:
    def __init__(self, model):
        self.model = model
        self.engine = model.engine
        self.is_running = False
        self.is_starting = False
        self.t = 0
        self.input_shape = None
        self.content_shape = None
        self.has_grams = False
        # reference worker.py:129-133: all-ones weights on every blob until SetWeights arrives
        self.rows = list(self.model.layers())
        self.cells = {k: OrderedDict((n, F32(1)) for n in self.rows) for k in LOSS_NAMES}
        self.params = {w: 1 for w in SCALAR_LOSS_NAMES}
        self._push_weights()
        self.optimizer = None
        self.optimizer_cls = LBFGSOptimizer                    # reference worker.py:135
        self.step_size = 1                                      # SetOptimizer.step_sizes['lbfgs']
        self.traces = []
        self._inflight = []

    # ------------------------------------------------------------------ views of device state
    @property
    def input(self):
        return self.engine.get_input_nchw() if self.input_shape is not None else None

    @property
    def content(self):
        return self.engine.get_content_nchw() if self.content_shape is not None else None

    @property
    def weights(self):
        return self.cells

    # ------------------------------------------------------------------ reference worker.py:140-152
    def check_consistency(self):
        return (self.input_shape is not None and self.content_shape is not None and self.has_grams
                and self.input_shape == self.content_shape)

    def objective_changed(self):
        if self.optimizer is not None:
            self.optimizer.objective_changed()

    def pause(self):
        self.is_running = False
        self.is_starting = False

    # ------------------------------------------------------------------ reference worker.py:154-170
    def resample_input(self, size):
        size = tuple(size)
        if self.input_shape is not None and self.optimizer is not None:
            self.optimizer.resample(size)
        else:
            self.engine.set_input_nchw(np.zeros((1, 3) + size, F32))
        self.input_shape = (1, 3) + size
        self._start()
        self.objective_changed()

    def resample_content(self, size):
        size = tuple(size)
        if self.content_shape is not None:
            self.engine.resample_content(size)                      # Lanczos on the device, then the features
        else:
            self.engine.set_content_nchw(np.zeros((1, 3) + size, F32))
        self.content_shape = (1, 3) + size
        self._start()
        self.objective_changed()

    # ------------------------------------------------------------------ reference worker.py:172-189
    def reset(self):
        self.engine.clear_norms()
        self.t = 0
        self.optimizer = self.optimizer_cls(self.engine, self.opfunc, step_size=self.step_size)

    def start(self):
        self.is_starting = True
        self._start()
        return self.is_running

    def _start(self):
        if self.is_starting and self.check_consistency():
            if self.optimizer is None:
                self.reset()
            self.is_starting = False
            self.is_running = True

    # ------------------------------------------------------------------ reference worker.py:191-229
    def set_input(self, image):
        shape = (1, 3) + tuple(np.shape(image)[:2])
        if self.input_shape is not None and self.input_shape == shape:
            self.engine.set_input(image)
            self.objective_changed()
        elif self.optimizer is not None:
            self.optimizer.resample(None, new_x=self.model.preprocess(image))
            self.input_shape = shape
            self._start()
        else:
            self.engine.set_input(image)
            self.input_shape = shape
            self.reset()
            self._start()

    def set_content(self, image):
        self.content_shape = (1, 3) + tuple(np.shape(image)[:2])
        self.engine.set_content(image)
        self._start()
        self.objective_changed()

    def set_style(self, image):
        self.engine.set_style(image)
        self.has_grams = True
        self._start()
        self.objective_changed()

    def set_step_size(self, step_size):
        self.step_size = step_size
        if self.optimizer is not None:
            self.optimizer.step_size = step_size

    def set_weights(self, weights, params):
        for kind in LOSS_NAMES:
            weights[kind]                                   # same KeyError as w['content'][layer]
        self.rows, self.cells = weight_table(weights)
        self.params = params
        self._push_weights()
        self.objective_changed()

    def _push_weights(self):
        cols = [[self.cells[k][r] for r in self.rows] for k in LOSS_NAMES]
        self.engine.set_weights(self.rows, cols[0], cols[1], cols[2],
                                [self.params[k] for k in SCALAR_LOSS_NAMES])

    # ------------------------------------------------------------------ reference worker.py:231-301
    def _active(self):
        def on(kind, layer):
            return abs(self.cells[kind][layer]) > EPS
        return [(n, on('content', n), on('style', n), on('deepdream', n)) for n in self.rows
                if on('content', n) or on('style', n) or on('deepdream', n)]

    def _make_trace(self, values, with_grad):
        t = Trace()
        active = self._active()
        for i, (layer, c, s, d) in enumerate(active):
            v = values[6 * i:6 * i + 6]
            for flag, tag, off in ((c, 'c', 0), (s, 's', 2), (d, 'd', 4)):
                if flag:
                    t('%s_%s_loss' % (layer, tag), v[off])
                    t('%s_%s_grad' % (layer, tag), v[off + 1])
        g = values[6 * len(active):]
        t('scd_loss', g[0])
        t('t_loss', g[1])
        t('p_loss', g[2])
        if with_grad:
            t('scd_grad', g[3])
            t('t_grad', g[4])
            t('p_grad', g[5])
            t('time', time.perf_counter())
        t('loss', g[6])
        if with_grad:
            t('grad', g[7])
        return t

    def opfunc(self, x=None, return_grad=True):
        """Objective and gradient at the current input (or at ``x``, which then becomes the input)."""
        if x is not None:
            self.engine.set_input_nchw(x)
        loss, grad, values = self.engine.opfunc(return_grad)
        self.traces.append(self._make_trace(values, return_grad))
        return (loss, grad) if return_grad else loss

    # ------------------------------------------------------------------ reference worker.py:303-315
    def step(self):
        """Returns the next iterate (HxWx3 float32 RGB, unclipped) and the trace of the iteration."""
        self.t += 1
        image, values, _ = self.engine.step(want_image=True, want_trace=True)
        t = self._make_trace(values, True)
        t('fevals', self.t)
        self.traces.append(t)
        return image, t.data

    # The same iteration in two halves (st_step_begin / st_step_end): the worker loop begins iteration k + 1 before it collects
    # iterate k, so the GPU never waits for the copy to the host and the pickling.  Results are those of step(), bit for bit.
    def step_begin(self):
        self.engine.step_begin()                # (first: a refused begin must not advance the iteration count)
        self.t += 1
        self._inflight.append(self.t)

    def step_end(self, copy=True, frame=False):
        """(image, trace, iteration index) of the oldest iteration begun and not yet collected.  copy=False: the image is a
        read-only view that stays valid for the next Engine.STEP_VIEW_LIFETIME calls of step_begin.  frame=True (after
        enable_iterate_frames): a fourth result, the finished pickle of ``messages.Iterate(image, index, trace)`` as a memoryview
        of the same pinned buffer (iterate_frame.py) -- what a transport sends without copying the image on the host."""
        if frame:
            image, values, _, room = self.engine.step_end(False, room=True)
        else:
            image, values, _ = self.engine.step_end(copy)
        index = self._inflight.pop(0)
        t = self._make_trace(values, True)
        t('fevals', index)
        self.traces.append(t)
        if frame:
            from . import iterate_frame
            head_room = self.engine._frame_room[0]
            return image, t.data, index, iterate_frame.assemble(room, head_room, image.nbytes, image.shape, index, t.data)
        return image, t.data, index

    def enable_iterate_frames(self):
        """From the next step_begin on, iterates come with room for the pickle of their ``messages.Iterate`` around them."""
        from . import iterate_frame
        self.engine.set_frame_room(iterate_frame.HEAD_ROOM, iterate_frame.TAIL_ROOM)

    @property
    def steps_pending(self):
        return len(self._inflight)

    def step_async(self):
        """Device-resident iteration: nothing is read back, the call does not wait for the GPU."""
        self.t += 1
        self.engine.step(want_image=False, want_trace=False)

    def write_trace(self, filename):
        keys = []
        for t in self.traces:
            keys.extend(k for k in t.data if k not in keys)
        with open(filename, 'w', newline='') as f:
            out = csv.writer(f)
            out.writerow(['step'] + keys)
            for i, t in enumerate(self.traces):
                out.writerow([i] + [t.data.get(k, '') for k in keys])
