#!/usr/bin/env python3
"""Tile-sharded single-image benchmark (BASELINE config 5).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_tiled.py \
        --size 8192 --grid 2x4 --steps 10
    python tools/bench_tiled.py --size 2048 --grid 1x1          # one GPU: the tile phases without communication

One rank per GPU.  Default: the iteration with its communication inside the engine (st_tile_step: RCCL all-reduces and grouped
send / recv strip exchanges on the engine's stream; torch is not imported, the communicator id travels over a plain socket).
--driver phases: the phase-by-phase driver over torch.distributed (the path the gloo tests use).  One JSON line on rank 0."""
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
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--optimizer', default='adam', choices=('adam', 'lbfgs'))
ap.add_argument('--driver', default='engine', choices=('engine', 'phases'))
ap.add_argument('--solo-rank', type=int, default=-1,
                help='run ONE rank of the grid alone on this GPU (its window, its pack / unpack kernels; nothing is exchanged, what a\n'
                     'neighbour would send arrives as zeros): the per-rank compute of a multi-GPU run, e.g. --size 8192 --grid 2x4 --solo-rank 1\n'
                     '= an interior column (window 4176 x 2208)')
ap.add_argument('--precision', default='fp32', choices=('fp32', 'bf16', 'bf16-full'), help='bf16: conv operands bf16 (lean data flow), everything else fp32; bf16-full: every fp32 tensor written')
args = ap.parse_args()
# native libraries (RCCL prints its version banner, gloo, the HIP runtime) write to fd 1 at will: keep the real stdout for the ONE JSON line
sys.stdout.flush()
json_out = os.fdopen(os.dup(1), 'w')
os.dup2(2, 1)
rows, cols = (int(v) for v in args.grid.split('x'))
gH, gW = (int(v) for v in args.size.split('x')) if 'x' in args.size else (int(args.size), int(args.size))
rank, local, world = int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
solo = args.solo_rank >= 0
if not solo and world != rows * cols:       # preflight: a clear error instead of ranks waiting for peers that do not exist
    sys.exit('bench_tiled.py: the launcher started %d rank(s) but the %s grid needs %d (one rank per tile)' % (world, args.grid, rows * cols))
if solo:
    assert world == 1 and args.solo_rank < rows * cols
    rank = args.solo_rank
dist = None
use_engine = args.driver == 'engine'
if world > 1 and not use_engine:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))

import style_transfer2_amd as st2                                    # noqa: E402  (after torch, when torch is used at all)
from style_transfer2_amd import tiled, tiling, weights as st2_weights   # noqa: E402
from style_transfer2_amd.tile_backend import HipTileBackend             # noqa: E402

WEIGHTS = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1},
           'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
topo = st2.VGG19_TOPOLOGY
grid = tiling.TileGrid(gH, gW, rows, cols, topo, 17)
win = grid.windows[rank]


def window_image(seed):
    """Deterministic pseudo-random image, generated per window so that no rank holds the full 8192^2 image."""
    yy, xx = np.mgrid[win.y0:win.y1, win.x0:win.x1].astype(np.uint32)
    out = np.empty((win.y1 - win.y0, win.x1 - win.x0, 3), np.uint8)
    for c in range(3):
        hsh = (yy * np.uint32(73856093)) ^ (xx * np.uint32(19349663)) ^ np.uint32((seed * 3 + c) * 83492791)
        hsh ^= hsh >> np.uint32(13)
        hsh *= np.uint32(1274126177)
        out[..., c] = (hsh >> np.uint32(11)) & np.uint32(255)
    return out


class WindowView:
    """Stands in for a full image: HipTileBackend only ever crops its own window out of it."""
    def __init__(self, arr):
        self.arr = arr

    def __getitem__(self, idx):
        ys, xs = idx
        return self.arr[ys.start - win.y0:ys.stop - win.y0, xs.start - win.x0:xs.stop - win.x0]


style = np.random.RandomState(2).randint(0, 256, (args.style_size, args.style_size, 3)).astype(np.uint8)
backend = HipTileBackend(st2_weights.he_normal(topo, seed=0), grid, rank, WindowView(window_image(1)), style,
                         WindowView(window_image(3)), WEIGHTS, PARAMS, step_size={'adam': 10, 'lbfgs': 1}[args.optimizer], device=local,
                         precision=args.precision, use_torch=not use_engine, optimizer=args.optimizer)


class SoloComm(tiled.Comm):
    """One rank of the grid on its own: reductions see one rank, a neighbour's strips arrive as zeros."""
    def exchange(self, sends, recvs):
        for _, t in recvs:
            t.zero_()


if use_engine:
    import ctypes
    from style_transfer2_amd import capi

    def make_id():
        uid = ctypes.create_string_buffer(capi.COMM_ID_BYTES)
        capi.check(backend.lib.st_comm_unique_id(uid))
        return uid.raw
    if solo:
        backend.comm_init_solo(rank, rows * cols)
    else:
        backend.comm_init_rccl(tiled.rendezvous_unique_id(rank, world, make_id), rank, world)
    tt = tiled.FusedTiledTransfer(grid, rank, backend)
    barrier = (lambda: None) if solo else backend.barrier
else:
    tt = tiled.TiledTransfer(grid, rank, backend, SoloComm() if solo else tiled.Comm(dist, rank, world), optimizer=args.optimizer,
                             step_size={'adam': 10, 'lbfgs': 1}[args.optimizer])
    barrier = dist.barrier if dist is not None else (lambda: None)
for _ in range(args.warmup):
    tt.step()
backend.engine.sync()
barrier()
t0 = time.perf_counter()
for k in range(args.steps):
    if use_engine and k + 1 < args.steps:
        tt.step_async()                     # nothing read back: the host enqueues ahead of the GPU; the last step brings the trace
    else:
        vals = tt.step()
backend.engine.sync()
barrier()
dt = time.perf_counter() - t0
if rank == 0 or solo:
    json_out.write(json.dumps({'metric': 'tile-sharded style-transfer iters/sec @%sx%s VGG19' % (gH, gW), 'value': args.steps / dt,
                      'unit': 'it/s', 'n_gpus': world, 'grid': args.grid, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
                      'higher_is_better': True, 'vs_baseline': None,
                      'config': {'workload': 'configs[4]: ONE %dx%d image tile-sharded %s (apron design, RCCL all-reduces + strip exchange), %s %s' % (gH, gW, args.grid, args.optimizer, args.precision),
                                 'measured_on_hardware': 'by the driver only; the builder has one GPU'},
                      'driver': 'engine (st_tile_step, RCCL inside the engine)' if use_engine else 'phases (torch.distributed between the st_tile_* phases)',
                      'solo_rank': (args.solo_rank if solo else None), 'apron_px': grid.apron, 'window': [win.y1 - win.y0, win.x1 - win.x0], 'loss': float(vals[-2]),
                      'dtype': 'f32' if args.precision == 'fp32' else 'bf16 conv operands, f32 accumulate/Gram/optimizer', 'scaling': 'strong', 'data': 'synthetic'}) + '\n')
    json_out.flush()
if dist is not None:
    dist.destroy_process_group()
