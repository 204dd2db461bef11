"""The style-transfer objective and its driver, restated on the CPU.  TEST INFRASTRUCTURE.

Follows worker.py:109-114 (Gram matrix), worker.py:117-229 (state), worker.py:231-301
(``opfunc``), worker.py:303-310 (``step``) and utils.py:257-282 (``Trace``).
The model is injected (any object with layers/preprocess/deprocess/forward/backward, i.e.
``oracle.NetOracle`` or the HIP-backed model under test).
Pinned against the reference's own ``StyleTransfer`` by tests/golden/transfer_*.npz.
"""

from collections import OrderedDict
import time

import numpy as np

from . import descent, image_norms

F32 = np.float32
LOSS_KINDS = ('content', 'style', 'deepdream')          # messages.py:146
SCALAR_PARAMS = ('tv', 'tv_power', 'p', 'p_power')       # messages.py:147
EPS_W = 1e-15                                            # worker.py:234,249,258,271


def gram(feat):
    """F F^T / (C h w) for a (1, C, h, w) feature map (worker.py:109-114)."""
    n, c, h, w = feat.shape
    assert n == 1
    f = feat.reshape(c, h * w)
    return np.dot(f, f.T) / F32(f.size)


class TraceLog:
    """Ordered name -> python scalar record; duplicate names get '_' appended (utils.py:257-282)."""

    def __init__(self):
        self.data = OrderedDict()

    def put(self, name, value):
        while name in self.data:
            name += '_'
        if isinstance(value, np.floating):
            self.data[name] = float(value)
        elif isinstance(value, np.integer):
            self.data[name] = int(value)
        else:
            self.data[name] = value
        return value

    def put_rms(self, name, arr):
        self.put(name, np.sqrt(np.mean(arr**2)))
        return arr


def weight_table(weights):
    """Rows (layer order) and cells of ``pd.DataFrame.from_dict(weights, dtype=float32)``
    (worker.py:226-229) without pandas: rows are the union of the inner keys in order of first
    appearance, absent cells are NaN (which every |w| > 1e-15 test treats as zero)."""
    rows = []
    for kind in weights:
        for layer in weights[kind]:
            if layer not in rows:
                rows.append(layer)
    cells = {kind: {layer: F32(weights[kind].get(layer, np.nan)) for layer in rows}
             for kind in weights}
    return rows, cells


class TransferOracle:
    """CPU restatement of worker.py:117-315 ``StyleTransfer`` (resampling excluded: SURVEY 8f)."""

    optimizers = {'adam': descent.AdamOracle, 'lbfgs': descent.LBFGSOracle}
    default_steps = {'adam': 10, 'lbfgs': 1}                      # messages.py:119

    def __init__(self, model):
        self.model = model
        self.is_running = self.is_starting = False
        self.t = 0
        self.input = self.content = self.features = self.grams = None
        names = model.layers()
        # worker.py:129-133: before any SetWeights every blob has weight 1 for every loss
        self.rows = list(names)
        self.cells = {k: {n: F32(1) for n in names} for k in LOSS_KINDS}
        self.params = {k: 1 for k in SCALAR_PARAMS}
        self.optimizer = None
        self.optimizer_kind = 'lbfgs'                              # worker.py:135-136
        self.step_size = self.default_steps['lbfgs']
        self.norms = {k: {} for k in 'cds'}
        self.traces = []
        # The reference keeps the content features and style Grams of EVERY blob (worker.py:204-216: the weights may change later),
        # which at full size is a whole-net forward and 22 blob copies per image.  A test that fixes its weights beforehand may name
        # the blobs it will weight: only those are then computed and kept -- the values are the same, the others are never read.
        self.feature_layers = None

    # --- state (worker.py:140-229) -----------------------------------------------------------
    def check_consistency(self):
        return (self.input is not None and self.content is not None and bool(self.grams)
                and self.input.shape == self.content.shape)

    def objective_changed(self):
        if self.optimizer is not None:
            self.optimizer.objective_changed()

    def pause(self):
        self.is_running = self.is_starting = False

    def reset(self):
        self.norms = {k: {} for k in 'cds'}
        self.t = 0
        cls = self.optimizers[self.optimizer_kind]
        self.optimizer = cls(self.input, self.opfunc, step_size=self.step_size)

    def start(self):
        self.is_starting = True
        self._maybe_start()
        return self.is_running

    def _maybe_start(self):
        if self.is_starting and self.check_consistency():
            if self.optimizer is None:
                self.reset()
            self.is_starting, self.is_running = False, True

    def set_input(self, image):
        x = self.model.preprocess(image)
        if self.input is not None and self.input.shape == x.shape:
            self.input[:] = x
            self.objective_changed()
        elif self.optimizer is not None:
            raise NotImplementedError('optimizer.resample path is out of the oracle scope')
        else:
            self.input = x
            self.reset()
            self._maybe_start()

    def set_content(self, image):
        self.content = self.model.preprocess(image)
        self.features = {k: v.copy() for k, v in self.model.forward(self.content, self.feature_layers).items()}
        self._maybe_start()
        self.objective_changed()

    def set_style(self, image):
        feats = self.model.forward(self.model.preprocess(image), self.feature_layers)
        self.grams = {k: gram(v) for k, v in feats.items()}
        self._maybe_start()
        self.objective_changed()

    def set_optimizer(self, kind, step_size=None):
        """worker.py:387-391 (SetOptimizer handling)."""
        self.optimizer_kind = kind
        self.step_size = step_size if step_size else self.default_steps[kind]
        if self.optimizer is not None:
            self.optimizer.step_size = self.step_size
        if not isinstance(self.optimizer, self.optimizers[kind]):
            self.reset()

    def set_weights(self, weights, params):
        self.rows, self.cells = weight_table(weights)
        self.params = params
        self.objective_changed()

    # --- objective (worker.py:231-301) -------------------------------------------------------
    def active_layers(self):
        def on(kind, layer):
            return abs(self.cells[kind][layer]) > EPS_W
        return [n for n in self.rows if any(on(k, n) for k in self.cells)]

    def opfunc(self, x, return_grad=True):
        log = TraceLog()
        layers = self.active_layers()
        feats = self.model.forward(x, layers)
        cn, sn, dn = (self.norms[k] for k in 'csd')
        loss = 0
        diffs = {}
        for layer in layers:
            cw, sw, dw = (self.cells[k][layer] for k in LOSS_KINDS)
            feat = feats[layer]
            acc = np.zeros_like(feat)

            if abs(cw) > EPS_W:                                   # worker.py:249-256
                d = feat - self.features[layer]
                g = (2 / d.size) * d
                if layer not in cn:
                    cn[layer] = np.sqrt(np.mean(g**2))
                loss += log.put(layer + '_c_loss', cw * np.mean(d**2) / cn[layer])
                acc += log.put_rms(layer + '_c_grad', cw * g / cn[layer])

            if abs(sw) > EPS_W:                                   # worker.py:258-269
                _, c, mh, mw = feat.shape
                # bf16 feature path emulation: the Gram of the CURRENT features is the fp32-accumulated product of their
                # bf16-rounded values where the engine takes it from the bf16 copy (gram16.hip); the style targets stay fp32
                gop = self.model.gram_operand(layer, feat) if hasattr(self.model, 'gram_operand') else feat
                gd = gram(gop) - self.grams[layer]
                f2 = feat.reshape(c, mh * mw)
                # bf16 feature path emulation (BASELINE config 3): the product D @ F takes the bf16-rounded features of a
                # conv blob with C % 64 == 0 (the engine's style16.hip); D and everything else stay fp32
                fop = self.model.style_operand(layer, f2) if hasattr(self.model, 'style_operand') else f2
                sg = np.dot(gd, fop).reshape(1, c, mh, mw)
                sg *= 2 / (gd.size * f2.size)
                if layer not in sn:
                    sn[layer] = np.sqrt(np.mean(sg**2))
                loss += log.put(layer + '_s_loss', sw * np.mean(gd**2) / sn[layer])
                log.put_rms(layer + '_s_grad', sw / sn[layer] * sg)
                descent.saxpy(sw / sn[layer], sg, acc)

            if abs(dw) > EPS_W:                                   # worker.py:271-277
                g = (-2 / feat.size) * feat
                if layer not in dn:
                    dn[layer] = np.sqrt(np.mean(g**2))
                loss += log.put(layer + '_d_loss', -dw * np.mean(feat**2) / dn[layer])
                acc += log.put_rms(layer + '_d_grad', dw * g / dn[layer])

            diffs[layer] = acc

        log.put('scd_loss', loss)
        tv_value, tv_grad = image_norms.tv_term(x / 255, self.params['tv_power'])
        loss += log.put('t_loss', self.params['tv'] * tv_value)
        p_value, p_grad = image_norms.p_term(x / 255, self.params['p_power'])
        loss += log.put('p_loss', self.params['p'] * p_value)

        if not return_grad:
            self.traces.append(log)
            return log.put('loss', loss)

        grad = log.put_rms('scd_grad', self.model.backward(diffs).copy())
        grad += log.put_rms('t_grad', self.params['tv'] * tv_grad)
        grad += log.put_rms('p_grad', self.params['p'] * p_grad)
        log.put('time', time.perf_counter())
        self.traces.append(log)
        return log.put('loss', loss), log.put_rms('grad', grad)

    # worker.py:303-310
    def step(self):
        self.t += 1
        x, _ = self.optimizer.step()
        log = self.traces[-1]
        log.put('fevals', self.t)
        return self.model.deprocess(x), log.data
