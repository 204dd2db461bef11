#!/usr/bin/env python3
"""bf16 path at 2048^2: objective gradient with the style term fused into the data-gradient convs against the separate kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
out = {}
for flag in ('1', '0'):
    os.environ['ST2_STYLE_FUSE'] = flag
    job = bench.make_job(bench.images(2048) + (bench.WEIGHTS, bench.PARAMS), 'lbfgs', 0, 'bf16')
    job.opfunc()
    loss, grad = job.opfunc()
    out[flag] = (loss, grad.astype(np.float64).copy(), dict(job.traces[-1].data))
    del job
(l1, g1, t1), (l0, g0, t0) = out['1'], out['0']
rel = np.linalg.norm(g1 - g0) / np.linalg.norm(g0)
cos = float(np.vdot(g1, g0) / (np.linalg.norm(g1) * np.linalg.norm(g0)))
print('loss fused %r separate %r' % (l1, l0))
print('gradient rel-L2 %.3e cosine %.9f' % (rel, cos))
for k in t0:
    if k.endswith('_s_grad'):
        print('%-16s fused %.6e separate %.6e rel %.2e' % (k, t1[k], t0[k], abs(t1[k] - t0[k]) / abs(t0[k])))
