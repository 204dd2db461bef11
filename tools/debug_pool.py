import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oracle, style_transfer2_amd as st2
from helpers import rel_l2
rs = np.random.RandomState
for C in (16, 128):
    topo = (('conv', 'conv1_1', 3, C), ('pool', 'pool1'))
    params = oracle.he_init_weights(topo, seed=0)
    cpu = oracle.NetOracle(topo, params); gpu = st2.HipModel(params, topology=topo)
    for sz in ((112, 152), (224, 304), (96, 128), (56, 76), (113, 150)):
        x = (rs(7).randn(1, 3, *sz) * 40).astype(np.float32)
        fc = cpu.forward(x); fg = gpu.forward(x)
        d = rs(5).randn(*fc['pool1'].shape).astype(np.float32)
        for rep in range(2):
            gc = cpu.backward({'pool1': d}); gg = gpu.backward({'pool1': d})
            e = np.abs(gg - gc)[0].max(0)
            print('C=%d size %s fwd pool rel %.1e | bwd rel %.2e worst %s frac_bad %.5f' % (C, sz, rel_l2(fg['pool1'], fc['pool1']), rel_l2(gg, gc), np.unravel_index(e.argmax(), e.shape), np.mean(e > 1e-3 * np.abs(gc).max())))
