#!/usr/bin/env python3
"""Headless stylisation job on one MI355X:  stylize.py content.jpg style.jpg out.png [--size 512] [--iters 500]
--grid RxC: an image too large for one engine (beyond about 4096 x 4096) is cut into R x C tiles that all live on the one GPU
(jobs.run_tiled_job; the image edges must divide into tiles of multiples of 16 pixels)."""
import argparse
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import style_transfer2_amd as st2                                   # noqa: E402
from style_transfer2_amd import jobs, weights as st2_weights         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('content'); ap.add_argument('style'); ap.add_argument('out')
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--style-size', type=int, default=0)
ap.add_argument('--iters', type=int, default=500)
ap.add_argument('--optimizer', default='adam', choices=['adam', 'lbfgs'])
ap.add_argument('--weights', default='', help='.npz or .caffemodel; default: seeded synthetic weights')
ap.add_argument('--gpu', type=int, default=0)
ap.add_argument('--grid', default='', help='RxC: tile-shard the image over this one GPU (large images)')
args = ap.parse_args()

if args.weights.endswith('.npz'):
    params = st2_weights.load_npz(args.weights, st2.VGG19_TOPOLOGY)
elif args.weights.endswith('.caffemodel'):
    from style_transfer2_amd import caffemodel
    params = caffemodel.vgg_params(caffemodel.read_caffemodel(args.weights), st2.VGG19_TOPOLOGY)
else:
    params = st2_weights.he_normal(st2.VGG19_TOPOLOGY, seed=0)
if args.grid:
    rows, cols = (int(v) for v in args.grid.split('x'))
    image = jobs.run_tiled_job(params, jobs.load_rgb(args.content), jobs.load_rgb(args.style), args.iters, (rows, cols), size=args.size,
                               style_size=args.style_size or None, device=args.gpu, optimizer=args.optimizer)
else:
    job = st2.StyleTransfer(st2.HipModel(params, device=args.gpu))
    image = jobs.run_job(job, jobs.load_rgb(args.content), jobs.load_rgb(args.style), args.iters, size=args.size,
                         style_size=args.style_size or None, optimizer=args.optimizer)
Image.fromarray(np.uint8(np.clip(image, 0, 255))).save(args.out)
print('wrote', args.out, image.shape)
