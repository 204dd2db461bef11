import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oracle, style_transfer2_amd as st2
from helpers import rel_l2
rs = np.random.RandomState
topo = oracle.VGG19_TOPOLOGY[:11]     # through pool3
params = oracle.he_init_weights(topo, seed=0)
cpu = oracle.NetOracle(topo, params); gpu = st2.HipModel(params, topology=topo)
for sz in ((224, 304), (225, 300), (96, 128)):
    x = (rs(7).randn(1, 3, *sz) * 40).astype(np.float32)
    fc = cpu.forward(x); fg = gpu.forward(x)
    print('size', sz, 'fwd worst rel %.2e' % max(rel_l2(fg[n], fc[n]) for n in fc))
    for layer in ['conv1_1', 'conv1_2', 'pool1', 'conv2_1', 'conv2_2', 'pool2', 'conv3_1', 'conv3_4', 'pool3']:
        d = rs(5).randn(*fc[layer].shape).astype(np.float32)
        cpu.forward(x); gpu.forward(x)
        gc = cpu.backward({layer: d}); gg = gpu.backward({layer: d})
        e = np.abs(gg - gc)[0].max(0)
        print('   from %-8s rel %.2e  worst px %s  frac px > 1e-3*max: %.4f' % (layer, rel_l2(gg, gc), np.unravel_index(e.argmax(), e.shape), np.mean(e > 1e-3 * np.abs(gc).max())))

print('--- same test, CPU backward using the GPU forward blobs for ReLU masks / pool arg-max')
from oracle import caffe_net as cn
for sz in ((224, 304), (225, 300)):
    x = (rs(7).randn(1, 3, *sz) * 40).astype(np.float32)
    fc = cpu.forward(x); fg = gpu.forward(x)
    flips = {n: int(np.sum((fc[n] > 0) != (fg[n] > 0))) for n in fc if n.startswith('conv')}
    print('size', sz, 'ReLU sign flips per blob:', {k: v for k, v in flips.items() if v})
    for n in fc:                                   # overwrite the oracle's saved forward state with the GPU's
        cpu._blobs[n] = fg[n][0].copy()
    names = cpu.blob_names
    for i, layer in enumerate(cpu.topology):
        if layer[0] == 'pool':
            _, slot_gpu = cn.maxpool_forward(cpu._blobs[names[i]])
            print('   %s arg-max flips: %d' % (layer[1], int(np.sum(slot_gpu != cpu._slots[layer[1]]))))
            cpu._slots[layer[1]] = slot_gpu
    for layer in ['pool2', 'conv3_4', 'pool3']:
        d = rs(5).randn(*fc[layer].shape).astype(np.float32)
        gc = cpu.backward({layer: d}); gg = gpu.backward({layer: d})
        print('   from %-8s rel %.2e' % (layer, rel_l2(gg, gc)))
