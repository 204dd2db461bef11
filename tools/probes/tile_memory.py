import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
torch.cuda.init()
import oracle
from style_transfer2_amd import tiled, tiling
from style_transfer2_amd.tile_backend import HipTileBackend
prec = sys.argv[1]
h = w = 8192
rs = np.random.RandomState
content = rs(1).randint(0, 256, (h, w, 3)).astype(np.uint8); init = rs(3).randint(0, 256, (h, w, 3)).astype(np.uint8)
style = rs(2).randint(0, 256, (96, 80, 3)).astype(np.uint8)
weights = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1}, 'deepdream': {}}
params = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
net = oracle.he_init_weights(oracle.VGG19_TOPOLOGY, seed=0)
grid = tiling.TileGrid(h, w, 2, 4, oracle.VGG19_TOPOLOGY, 17)
free0, total = torch.cuda.mem_get_info()
b = HipTileBackend(net, grid, 1, content, style, init, weights, params, step_size=10, precision=prec)
b.comm_init_solo(1, 8)
tt = tiled.FusedTiledTransfer(grid, 1, b)
free1, _ = torch.cuda.mem_get_info()
t = time.time(); tt.step(); tt.step(); b.engine.sync(); dt = time.time() - t
free2, _ = torch.cuda.mem_get_info()
print(prec, 'window', grid.windows[1], 'after build %.2f GB, after 2 steps %.2f GB of %.0f GB; 2 steps %.3f s' % ((free0 - free1) / 2**30, (free0 - free2) / 2**30, total / 2**30, dt))
