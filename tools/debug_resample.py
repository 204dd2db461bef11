import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle, style_transfer2_amd as st2
from style_transfer2_amd import resample
topo = oracle.tiny_topology((8, 16), (2, 2))
eng = st2.Engine(topo); eng.load_weights(oracle.he_init_weights(topo, 0, 0.1))
x = (np.random.RandomState(0).randn(1, 3, 32, 40) * 50).astype(np.float32)
eng.set_input_nchw(x)
eng.optimizer_reset(2, 1.0)
for size in ((48, 64), (32, 64), (48, 40), (20, 26)):
    eng.set_input_nchw(x)
    want = resample.resample_nchw(x, size)
    ref2 = resample.resample_planes_reference(x, size)
    eng.resample_state(size)
    got = eng.get_input_nchw()
    d = np.abs(got - want)
    print(size, 'pillow==numpy', np.array_equal(want, ref2), 'max abs', d.max(), 'n diff', int((d > 0).sum()), 'of', d.size,
          'max rel', (d / np.maximum(np.abs(want), 1e-30)).max())
