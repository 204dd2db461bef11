#!/usr/bin/env python3
"""What do the ReLU-mask and injected-diff loads of the data-gradient epilogue cost?  Times the step's masked Winograd data-gradient
shapes at 1024^2 with both / mask only / inject only / neither (tools/probes st_bench_conv): the difference to "neither" bounds from
above what any cheaper mask representation (one bit per element) could give back.  GPU box only."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import probes
lib = probes.load_library()
SHAPES = [('conv1_2 dgrad (h4)', 64, 64, 1024, 109), ('conv2_2 dgrad', 128, 128, 512, 101), ('conv3_x dgrad', 256, 256, 256, 101),
          ('conv4_x dgrad', 512, 512, 128, 101)]
NAMES = {1: 'mask+inject', 2: 'mask', 3: 'inject', 4: 'neither', 0: 'forward'}
for name, K, M, edge, cfg in SHAPES:
    row = []
    for mode in (1, 2, 3, 4, 0):
        best = 1e9
        for _ in range(3):
            ms, used = ctypes.c_double(), ctypes.c_int()
            rc = lib.st_bench_conv(0, K, M, edge, edge, cfg, mode, 20, ctypes.byref(ms), ctypes.byref(used))
            assert rc == 0, lib.st_probe_last_error()
            best = min(best, ms.value)
        row.append('%s %.1f us' % (NAMES[mode], 1e3 * best))
    print('%-20s K=%d M=%d %dx%d: %s' % (name, K, M, edge, edge, ' | '.join(row)), flush=True)
