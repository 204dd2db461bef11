import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oracle, style_transfer2_amd as st2
from helpers import rel_l2
topo = oracle.VGG19_TOPOLOGY
params = oracle.he_init_weights(topo, seed=0)
rs = np.random.RandomState
H, W = 225, 300
content = rs(1).randint(0, 256, (H, W, 3)).astype(np.uint8)
style = rs(2).randint(0, 256, (187, 300, 3)).astype(np.uint8)
init = rs(3).randint(0, 256, (H, W, 3)).astype(np.uint8)
P4 = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
cases = {
 'base': {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1}, 'deepdream': {}},
 'content_pool3': {'content': {'pool3': 0.01}, 'style': {}, 'deepdream': {}},
 'style_pool4': {'content': {}, 'style': {'pool4': 0.5}, 'deepdream': {}},
 'dd_conv5_1': {'content': {}, 'style': {}, 'deepdream': {'conv5_1': 0.01}},
 'style_conv5_1': {'content': {}, 'style': {'conv5_1': 1}, 'deepdream': {}},
 'content_conv4_2': {'content': {'conv4_2': 0.08}, 'style': {}, 'deepdream': {}},
}
cpu_net = oracle.NetOracle(topo, params, full_forward=False)
gpu_model = st2.HipModel(params)
# forward parity per blob at this size
x = cpu_net.preprocess(init)
fc = cpu_net.forward(x); fg = gpu_model.forward(x)
for n in fc:
    print('fwd %-8s %s rel %.2e' % (n, fc[n].shape[1:], rel_l2(fg[n], fc[n])))
for name, w in cases.items():
    cpu = oracle.TransferOracle(cpu_net); dev = st2.StyleTransfer(gpu_model)
    for st in (cpu, dev):
        st.set_input(init); st.set_content(content); st.set_style(style); st.reset(); st.set_weights(w, P4)
    lo, go = cpu.opfunc(cpu.input); ld, gd = dev.opfunc()
    d = (gd - go)[0]
    e = np.sqrt((d**2).sum(0)); r = np.sqrt((go[0]**2).sum(0))
    print('%-16s grad rel %.2e loss rel %.2e | interior rel %.2e | worst px %s val %.3g' % (
        name, rel_l2(gd, go), abs(ld-lo)/abs(lo), rel_l2(gd[..., 16:-16, 16:-16], go[..., 16:-16, 16:-16]),
        np.unravel_index(e.argmax(), e.shape), e.max()))
    tc, td = cpu.traces[-1].data, dev.traces[-1].data
    for k in tc:
        if k != 'time' and abs(td[k]-tc[k]) > 2e-4*abs(tc[k]):
            print('    trace', k, tc[k], td[k])

print('--- B2 seam: oracle objective over HipModel (GPU forward/backward only)')
for name in ('style_conv5_1', 'style_pool4', 'content_conv4_2'):
    w = cases[name]
    cpu = oracle.TransferOracle(cpu_net); mix = oracle.TransferOracle(gpu_model)
    for st in (cpu, mix):
        st.set_input(init); st.set_content(content); st.set_style(style); st.reset(); st.set_weights(w, P4)
    lo, go = cpu.opfunc(cpu.input); lm, gm = mix.opfunc(mix.input)
    print('%-16s grad rel %.2e' % (name, rel_l2(gm, go)))
    # feed identical diffs to both backward implementations
    feats = cpu_net.forward(x, ['conv5_1'])
    d = np.random.RandomState(5).randn(*feats['conv5_1'].shape).astype(np.float32) * 1e-4
    gpu_model.forward(x, ['conv5_1'])
    print('   backward(same random diff at conv5_1) rel %.2e' % rel_l2(gpu_model.backward({'conv5_1': d}), cpu_net.backward({'conv5_1': d})))
for sz in ((96, 128), (225, 300), (224, 304)):
    xs = (rs(7).randn(1, 3, *sz) * 40).astype(np.float32)
    f = cpu_net.forward(xs, ['conv5_1', 'conv4_2', 'conv3_1']); gpu_model.forward(xs, ['conv5_1'])
    for layer in ('conv5_1', 'conv4_2', 'conv3_1'):
        d = rs(5).randn(*f[layer].shape).astype(np.float32)
        cpu_net.forward(xs, [layer]); gpu_model.forward(xs, [layer])
        print('size %s backward from %s rel %.2e' % (sz, layer, rel_l2(gpu_model.backward({layer: d}), cpu_net.backward({layer: d}))))
