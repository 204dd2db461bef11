"""Host handles of the two device-resident optimizers (reference optimizers.py:7-125).

The state (x, m, v / the L-BFGS pairs) lives in HBM inside the engine; these objects keep the
reference's interface: ``cls(x, opfunc, step_size=)``, ``.step()``, ``.resample(size, new_x)``,
``.objective_changed()`` and a ``.step_size`` attribute."""

from .engine import OPT_ADAM, OPT_LBFGS


class _DeviceOptimizer:
    kind = None

    def __init__(self, engine, opfunc=None, step_size=1):
        self.engine = engine
        self.opfunc = opfunc            # kept for interface parity; the objective is evaluated on device
        self._step_size = step_size
        engine.optimizer_reset(self.kind, step_size)

    @property
    def step_size(self):
        return self._step_size

    @step_size.setter
    def step_size(self, value):
        self._step_size = value
        self.engine.optimizer_set_step(value)

    def objective_changed(self):
        self.engine.objective_changed()


class AdamOptimizer(_DeviceOptimizer):
    """reference optimizers.py:7-46"""
    kind = OPT_ADAM

    def resample(self, size, new_x=None):
        """reference optimizers.py:29-40: x Lanczos (or replaced), m Lanczos, v bilinear clipped at 0 -- all on the
        device with Pillow's own coefficient tables (resample.pillow_coeffs), bit-exact with the reference's Pillow."""
        self.engine.resample_state(size, new_x)
        return None


class LBFGSOptimizer(_DeviceOptimizer):
    """reference optimizers.py:49-125 (n_corr = 10, fixed step, no line search)"""
    kind = OPT_LBFGS

    def resample(self, size, new_x=None):
        """reference optimizers.py:110-119"""
        self.engine.resample_state(size, new_x)
        self.objective_changed()
        return None
