#!/usr/bin/env python3
"""ONE image too large for one engine, tile-sharded over ONE GPU: every rank of the grid is an engine context of this process
(one thread each, time-sliced by the GPU), the all-reduces and strip exchanges are device-to-device copies
(style_transfer2_amd/tiled.py::InProcessFabric, HipTileBackend.comm_init_local).  The BASELINE configs[4] job end to end on the
hardware the builder has -- and the way a single MI355X runs an 8192^2 image at all.

    python tools/bench_tiled_one_gpu.py --size 8192 --grid 2x4 --steps 5        # fp32: eight windows of 24.7 GB
One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--size', default='8192', help='N or HxW of the whole image')
ap.add_argument('--style-size', type=int, default=1024)
ap.add_argument('--grid', default='2x4')
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--precision', default='fp32', choices=('fp32', 'bf16'))
ap.add_argument('--transport', default='device', choices=('device', 'host'))
ap.add_argument('--optimizer', default='adam', choices=('adam', 'lbfgs'), help='lbfgs: the fused Gram-form step (one all-reduce of the new inner products per step)')
ap.add_argument('--driver', default='engine', choices=('engine', 'phases'), help='phases: tiled.TiledTransfer between the st_tile_* phases (chain-form L-BFGS with scalar all-reduces)')
ap.add_argument('--need-free-gib', type=float, default=0.0, help='print {"skipped": ...} and exit 0 unless this much HBM is free')
args = ap.parse_args()
sys.stdout.flush()
json_out = os.fdopen(os.dup(1), 'w')
os.dup2(2, 1)
import torch                                                             # noqa: E402  (torch's HIP runtime first)
torch.cuda.init()
import style_transfer2_amd as st2                                        # noqa: E402
from style_transfer2_amd import tiled, tiling, weights as st2_weights    # noqa: E402
from style_transfer2_amd.tile_backend import HipTileBackend              # noqa: E402

if args.need_free_gib and torch.cuda.mem_get_info()[0] / 2 ** 30 < args.need_free_gib:
    json_out.write(json.dumps({'skipped': 'only %.0f GiB of HBM free, the eight resident windows need %.0f' % (torch.cuda.mem_get_info()[0] / 2 ** 30, args.need_free_gib)}) + '\n')
    json_out.flush()
    sys.exit(0)
rows, cols = (int(v) for v in args.grid.split('x'))
gH, gW = (int(v) for v in args.size.split('x')) if 'x' in args.size else (int(args.size), int(args.size))
WEIGHTS = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1},
           'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
topo = st2.VGG19_TOPOLOGY
grid = tiling.TileGrid(gH, gW, rows, cols, topo, 17)
world = rows * cols
rs = np.random.RandomState
content = rs(1).randint(0, 256, (gH, gW, 3)).astype(np.uint8)
init = rs(3).randint(0, 256, (gH, gW, 3)).astype(np.uint8)
style = rs(2).randint(0, 256, (args.style_size, args.style_size, 3)).astype(np.uint8)
net = st2_weights.he_normal(topo, seed=0)
free0, total = torch.cuda.mem_get_info()
fabric = tiled.InProcessFabric(world, timeout=600.0)
ranks, backends = [], []
for r in range(world):
    b = HipTileBackend(net, grid, r, content, style, init, WEIGHTS, PARAMS, step_size={'adam': 10, 'lbfgs': 1}[args.optimizer],
                       precision=args.precision, optimizer=args.optimizer)
    backends.append(b)
    if args.driver == 'phases':                  # the phase-by-phase driver (torch tensors between the st_tile_* phases): the A/B reference
        ranks.append(tiled.TiledTransfer(grid, r, b, tiled.LocalComm(fabric, r), optimizer=args.optimizer, step_size={'adam': 10, 'lbfgs': 1}[args.optimizer]))
        continue
    if args.transport == 'device':
        b.comm_init_local(r, world, fabric)
    else:
        b.comm_init_host(r, world, lambda v, r=r: fabric.allreduce(r, v), lambda s, rc, r=r: fabric.exchange(r, s, rc))
    ranks.append(tiled.FusedTiledTransfer(grid, r, b))
tiled.run_in_process(ranks, args.warmup, fabric)
for b in backends:
    b.engine.sync()
t0 = time.perf_counter()
out = tiled.run_in_process(ranks, args.steps, fabric)
for b in backends:
    b.engine.sync()
dt = time.perf_counter() - t0
used = (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30
json_out.write(json.dumps({
    'metric': 'tile-sharded style-transfer iters/sec @%sx%s VGG19, every rank on ONE GPU' % (gH, gW), 'value': args.steps / dt, 'unit': 'it/s',
    'n_gpus': 1, 'ranks': world, 'grid': args.grid, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
    'higher_is_better': True, 'vs_baseline': None,
    'config': {'workload': 'configs[4] job on one GPU: ONE %dx%d image, %s windows %s resident together, %s %s' % (
        gH, gW, args.grid, sorted({(w.y1 - w.y0, w.x1 - w.x0) for w in grid.windows}), args.optimizer, args.precision),
               'transport': 'in-process, %s' % ('device-to-device copies' if args.transport == 'device' else 'staged through host arrays'),
               'driver': 'st_tile_step (fused)' if args.driver == 'engine' else 'phase by phase (tiled.TiledTransfer)'},
    'hbm_in_use_GiB': used, 'loss': float(out[0][-1][-2]), 'all_reduces_per_step': fabric.reduces / max(1, args.steps + args.warmup),
    'messages_per_step': fabric.messages / max(1, args.steps + args.warmup),
    'dtype': 'f32' if args.precision == 'fp32' else 'bf16 conv operands, f32 accumulate/Gram/optimizer', 'data': 'synthetic'}) + '\n')
json_out.flush()
