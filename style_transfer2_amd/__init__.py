"""MI355X-native style-transfer inner loop (gfx950 HIP kernels behind a C ABI).

Host-side mirror of the reference's hot-path interfaces:
  * ``HipModel``       -- duck type of ``CaffeModel``            (reference worker.py:32-106)
  * ``StyleTransfer``  -- same state machine and method names     (reference worker.py:117-315)
  * ``AdamOptimizer`` / ``LBFGSOptimizer`` -- device-resident      (reference optimizers.py:7-125)

There is no CPU fallback: importing works anywhere, but creating an ``Engine`` without the built
HIP library or without a GPU raises ``HipUnavailable``.
"""

from .capi import HipUnavailable, StError, lib_path, load_library           # noqa: F401
from .engine import Engine, VGG19_TOPOLOGY                                   # noqa: F401
from .model import HipModel                                                  # noqa: F401
from .device_optimizers import AdamOptimizer, LBFGSOptimizer                 # noqa: F401
from .transfer import StyleTransfer, weight_table                            # noqa: F401
