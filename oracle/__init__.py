"""CPU oracle for the style-transfer inner loop -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (numpy + a small C file) of the
hot path of crowsonkb/style_transfer2 (``worker.py:32-315``, ``optimizers.py:7-125``,
``utils.py:29-69,232-304``, ``models/vgg19.prototxt``).  It exists so that the HIP
path can be checked against something that runs anywhere.

Rules (enforced by tests/test_boundary.py::test_product_code_never_imports_the_oracle):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it -- and only as the checker, never as the product;
  * nothing under ``style_transfer2_amd/``, ``worker.py`` or ``messages.py`` imports it;
  * the product path has no CPU fallback: it fails loudly without the HIP library.

Pinning status
  * everything above the network (Gram, content/style/deep-dream terms, TV, p-norm,
    trace, Adam, L-BFGS, state machine) is pinned against the reference itself:
    ``tests/golden/make_golden.py`` imports the reference's own ``worker.StyleTransfer``,
    ``optimizers`` and ``utils`` in the build container and stores their outputs as
    ``tests/golden/*.npz`` (the reference never travels; the fixtures do);
  * the network arithmetic (Caffe conv / in-place ReLU / ceil-mode max-pool and the
    ranged backward of ``worker.py:88-106``) has NO runnable reference here (pycaffe
    and the weights are absent, the reference has no tests): **parity unpinned at the
    Caffe boundary**.  It is pinned instead against torch CPU conv2d / max_pool2d
    (ceil_mode) / conv2d_input and hand-computed cases (tests/test_oracle_net.py).
"""

from .caffe_net import NetOracle, VGG19_TOPOLOGY, tiny_topology, he_init_weights  # noqa: F401
from .objective import TransferOracle, gram  # noqa: F401
from .image_norms import tv_term, p_term  # noqa: F401
from .descent import AdamOracle, LBFGSOracle, EmaBiasCorrected  # noqa: F401


def keep_freed_memory():
    """Tell glibc malloc to serve every request from the heap and never to give freed heap back to the kernel (mallopt
    M_MMAP_MAX = 0, M_TRIM_THRESHOLD = max) for the rest of this process.  The restated numpy code allocates a fresh temporary of
    blob size (hundreds of MB at 1024^2) in almost every line, exactly as the reference does; by default each of them is an mmap
    whose pages fault in one by one and are unmapped again a moment later -- about 40 % of an evaluation's wall time here.  This
    changes no arithmetic.  Called by tests/conftest.py and by bench.py's cpu_baseline leg, i.e. only where the oracle runs."""
    import ctypes
    try:
        libc = ctypes.CDLL('libc.so.6')
        return bool(libc.mallopt(-4, 0)) and bool(libc.mallopt(-1, 2 ** 31 - 1))
    except (OSError, AttributeError):
        return False
