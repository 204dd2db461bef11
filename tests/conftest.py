import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_device_contexts(request):
    """GPU tests build many engine contexts (each owns a HIP stream and device buffers).  StyleTransfer <-> optimizer
    reference cycles keep them alive until a garbage-collection pass, so collect after every GPU test: hardware queues
    are a finite per-process resource, and torch (used by the tile-sharded tests) brings up its own HIP runtime late."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        import gc
        gc.collect()


@pytest.fixture(scope='session', autouse=True)
def _torch_hip_runtime_first():
    """torch bundles its own copy of the HIP runtime; the engine library links /opt/rocm's.  Both live in one process in
    the tile-sharded tests (and in bench.py with N > 1, where torch.distributed comes up first).  Bring torch's runtime up
    FIRST here too -- the order bench.py uses and tools/coexist_check.py validates -- instead of whenever the first
    tile-sharded test happens to touch torch.cuda."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:           # no torch / no GPU: CPU-only session
        pass
    yield


@pytest.fixture(scope='session', autouse=True)
def _gpu_box_heartbeat():
    """The full-size parity tests spend minutes inside the CPU oracle without printing; the GPU box treats 7 silent
    minutes as a hang.  On a machine with a GPU, touch gpurun_out/.heartbeat once a minute while the session runs."""
    import threading
    import time
    stop = threading.Event()
    try:
        import torch
        on_gpu_box = torch.cuda.is_available()
    except Exception:
        on_gpu_box = False
    if on_gpu_box:
        out = os.path.join(REPO, 'gpurun_out')

        def beat():
            while not stop.wait(60.0):
                try:
                    os.makedirs(out, exist_ok=True)
                    with open(os.path.join(out, '.heartbeat'), 'w') as f:
                        f.write('%f\n' % time.time())
                except OSError:
                    pass
        threading.Thread(target=beat, name='heartbeat', daemon=True).start()
    yield
    stop.set()
