import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oracle, style_transfer2_amd as st2
from helpers import rel_l2
rs = np.random.RandomState
topo = (('conv', 'conv1_1', 3, 64), ('conv', 'conv1_2', 64, 128), ('conv', 'conv1_3', 128, 128))
params = oracle.he_init_weights(topo, seed=1, bias_std=0.1)
cpu = oracle.NetOracle(topo, params); gpu = st2.HipModel(params, topology=topo)
for sz in ((56, 76), (57, 75), (112, 152), (40, 200)):
    x = (rs(7).randn(1, 3, *sz) * 40).astype(np.float32)
    fc = cpu.forward(x); fg = gpu.forward(x)
    d = rs(5).randn(*fc['conv1_3'].shape).astype(np.float32)
    inj = rs(6).randn(*fc['conv1_2'].shape).astype(np.float32)
    gc = cpu.backward({'conv1_3': d, 'conv1_2': inj}); gg = gpu.backward({'conv1_3': d, 'conv1_2': inj})
    print('cfg=%s size %s fwd rel %.2e  bwd rel %.2e' % (os.environ.get('ST2_CONV_CFG', 'auto'), sz, rel_l2(fg['conv1_3'], fc['conv1_3']), rel_l2(gg, gc)), flush=True)
