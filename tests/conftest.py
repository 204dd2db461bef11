import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if os.environ.get('ST2_ORACLE_MALLOPT', '1') != '0':
        import oracle
        oracle.keep_freed_memory()      # the oracle's blob-sized numpy temporaries: reuse the heap instead of mmap / page-fault / munmap


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


# ---- progress beat for the GPU box -----------------------------------------------------------------------------------------
# The full-size parity tests spend minutes inside the CPU oracle without printing; the GPU box treats 7 silent minutes as a hang.
# The beat below is driven by PROGRESS only -- the start and end of every test, and every layer the oracle's network finishes
# (oracle.caffe_net.progress) -- never by a timer: a GPU call that hangs produces no further beat and the box's detector fires.
_last_beat = [0.0]


def _beat():
    import time
    now = time.time()
    if now - _last_beat[0] < 15.0:
        return
    _last_beat[0] = now
    try:
        out = os.path.join(REPO, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, '.heartbeat'), 'w') as f:
            f.write('%f\n' % now)
    except OSError:
        pass


def pytest_sessionstart(session):
    if os.environ.get('ST2_NO_HEARTBEAT'):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    from oracle import caffe_net
    caffe_net.progress = _beat


def pytest_runtest_logreport(report):
    from oracle import caffe_net
    if caffe_net.progress is not None:
        _last_beat[0] = 0.0             # a finished test phase always beats
        _beat()
