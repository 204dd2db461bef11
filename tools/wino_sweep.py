#!/usr/bin/env python3
"""Winograd vs direct conv3x3 kernel per VGG19 layer shape (GPU box).  TF/s are ALGORITHMIC (direct-conv flops)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import probes
lib = probes.load_library()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ITERS = int(os.environ.get('ITERS', '40'))
LAYERS = [('conv1_2', 64, 64, 1), ('conv2_1', 64, 128, 2), ('conv2_2', 128, 128, 2), ('conv3_1', 128, 256, 4),
          ('conv3_2', 256, 256, 4), ('conv4_1', 256, 512, 8), ('conv4_2', 512, 512, 8), ('conv5_1', 512, 512, 16)]
for name, cin, cout, div in LAYERS:
    hw = size // div
    for mode, (K, M) in (('fwd', (cin, cout)), ('bwd', (cout, cin))):
        fl = 2.0 * 9 * K * M * hw * hw
        out = []
        for cfg in (101, 102, 109, 100):
            ms = ctypes.c_double(); used = ctypes.c_int()
            rc = lib.st_bench_conv(0, K, M, hw, hw, cfg, 1 if mode == 'bwd' else 0, ITERS, ctypes.byref(ms), ctypes.byref(used))
            out.append('%7.3f ms %6.1f TF/s' % (ms.value, fl / ms.value / 1e9) if rc == 0 else '      (not eligible)   ')
        print('%-8s %s K=%-3d M=%-3d %4dx%-4d wino 128x(4x32): %s | 64x(8x32): %s | 64x(4x32) half, 2 WG/CU: %s | auto (+split-K): %s' % (name, mode, K, M, hw, hw, out[0], out[1], out[2], out[3]), flush=True)
