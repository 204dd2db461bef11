import os, sys
sys.path.insert(0, '.')
import numpy as np
import oracle, style_transfer2_amd as st2
rs = np.random.RandomState
size = 512
content, style, init = (rs(1).randint(0, 256, (size, size, 3)).astype(np.uint8), rs(2).randint(0, 256, (256, 256, 3)).astype(np.uint8), rs(3).randint(0, 256, (size, size, 3)).astype(np.uint8))
W = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1}, 'deepdream': {}}
P = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
params = oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0)
curves = {}
for prec in ('fp32', 'bf16'):
    for form in ('chain', 'gram'):
        os.environ['ST2_LBFGS_FORM'] = form
        job = st2.StyleTransfer(st2.HipModel(params, precision=prec))
        job.set_input(init); job.set_content(content); job.set_style(style); job.reset()
        job.set_weights(W, P); job.optimizer_cls = st2.LBFGSOptimizer; job.set_step_size(1); job.reset(); job.start()
        losses = [job.step()[1]['loss'] for _ in range(150)]
        curves[(prec, form)] = np.array(losses)
        print(prec, form, 'finite', bool(np.all(np.isfinite(losses))), 'first 6', ['%.6g' % v for v in losses[:6]], 'last', '%.6g' % losses[-1], 'min', '%.6g' % min(losses))
for prec in ('fp32', 'bf16'):
    a, b = curves[(prec, 'chain')], curves[(prec, 'gram')]
    rel = np.abs(a - b) / np.abs(a)
    print(prec, 'gram vs chain rel dev: steps 1-5 %.2e, 6-12 %.2e, 13-30 %.2e' % (rel[:5].max(), rel[5:12].max(), rel[12:30].max()))
