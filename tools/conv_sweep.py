#!/usr/bin/env python3
"""Isolated timing of the conv3x3 MFMA kernel per VGG19 layer shape and tile configuration (GPU box)."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import probes

lib = probes.load_library()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
only = sys.argv[2].split(',') if len(sys.argv) > 2 else None
LAYERS = [('conv1_1', 3, 64, 1), ('conv1_2', 64, 64, 1), ('conv2_1', 64, 128, 2), ('conv2_2', 128, 128, 2),
          ('conv3_1', 128, 256, 4), ('conv3_2', 256, 256, 4), ('conv4_1', 256, 512, 8), ('conv4_2', 512, 512, 8),
          ('conv5_1', 512, 512, 16)]
ITERS = int(os.environ.get("ITERS", "40"))
ncfg = lib.st_conv_num_configs()
names = [lib.st_conv_config_name(i).decode() for i in range(ncfg)]
print('configs:', names)
for name, cin, cout, div in LAYERS:
    if only and name not in only:
        continue
    hw = size // div
    for mode, (K, M) in (('fwd', (cin, cout)), ('bwd', (cout, cin))):
        if mode == 'bwd' and cin == 3:
            continue
        fl = 2.0 * 9 * K * M * hw * hw
        res = []
        used = ctypes.c_int()
        for cfg in [-1] + list(range(ncfg)):
            ms = ctypes.c_double()
            rc = lib.st_bench_conv(0, K, M, hw, hw, cfg, 1 if mode == "bwd" else 0, ITERS, ctypes.byref(ms), ctypes.byref(used))
            if rc != 0:
                res.append('   -  ')
                continue
            tag = '%5.1f' % (fl / ms.value / 1e9)
            if cfg == -1:
                tag += '(auto=%d)' % used.value
            res.append(tag)
        print('%-8s %s K=%-3d M=%-3d %4dx%-4d TF/s: %s' % (name, mode, K, M, hw, hw, '  '.join(res)), flush=True)
