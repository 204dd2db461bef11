#!/usr/bin/env python3
"""One rank of tests/test_gpu_multi_device.py: a FRESH process that owns ONE device (nothing is re-executed after GPU initialisation, no
torch): st_comm_unique_id on rank 0 -> TCP rendezvous -> st_comm_init(world) -> the tile grid 1 x world of one image -> fused Adam or
L-BFGS steps with every collective inside the engine over real RCCL (ncclAllReduce, grouped ncclSend / ncclRecv between devices).
usage: multi_device_child.py rank world port optimizer steps h w out.npz"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    rank, world, port, optimizer, steps, h, w, out = sys.argv[1:9]
    rank, world, port, steps, h, w = int(rank), int(world), int(port), int(steps), int(h), int(w)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['ST2_REQUIRE_PEER_ACCESS'] = os.environ.get('ST2_REQUIRE_PEER_ACCESS', '0')
    import oracle
    from style_transfer2_amd import capi, tiled, tiling
    from style_transfer2_amd.tile_backend import HipTileBackend
    rs = np.random.RandomState
    content, style, init = (rs(1).randint(0, 256, (h, w, 3)).astype(np.uint8), rs(2).randint(0, 256, (96, 80, 3)).astype(np.uint8),
                            rs(3).randint(0, 256, (h, w, 3)).astype(np.uint8))
    topo = oracle.VGG19_TOPOLOGY
    grid = tiling.TileGrid(h, w, 1, world, topo, 17)
    weights = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1}, 'deepdream': {}}
    backend = HipTileBackend(oracle.he_init_weights(topo, seed=0), grid, rank, content, style, init, weights,
                             {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}, step_size=10 if optimizer == 'adam' else 1,
                             device=rank, use_torch=False, optimizer=optimizer)

    def make_id():
        uid = ctypes.create_string_buffer(capi.COMM_ID_BYTES)
        capi.check(backend.lib.st_comm_unique_id(uid))
        return uid.raw
    uid = tiled.rendezvous_unique_id(rank, world, make_id, addr='127.0.0.1', port=port, timeout=120)
    backend.comm_init_rccl(uid, rank, world)
    ft = tiled.FusedTiledTransfer(grid, rank, backend)
    losses, grads, images = [], [], []
    for _ in range(steps):
        vals = ft.step()
        losses.append(vals[-2]); grads.append(vals[-1]); images.append(ft.tile_image())
    backend.barrier()
    np.savez(out, losses=np.array(losses), grads=np.array(grads), images=np.array(images), tile=np.array(tuple(grid.tiles[rank])))
    capi.check(backend.lib.st_comm_destroy(backend.ctx))
    return 0


if __name__ == '__main__':
    sys.exit(main())
