#!/usr/bin/env python3
"""Time the bf16 conv launcher on one shape with the options of one lean-flow launch: conv16_one.py K M H W mode[,mode...] [iters]
(modes: tools/probes/st2_probes.h, st_bench_conv16; ST2_CONV16_CFG / ST2_CONV16_SB_MAXK select the tile as in the product)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import probes
lib = probes.load_library()
K, M, H, W = [int(v) for v in sys.argv[1:5]]
modes = [int(v) for v in sys.argv[5].split(',')]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
for mode in modes:
    ms = ctypes.c_double()
    rc = lib.st_bench_conv16(0, K, M, H, W, mode, iters, ctypes.byref(ms))
    if rc:
        print('mode %d: rc=%d %s' % (mode, rc, lib.st_probe_last_error().decode()))
    else:
        print('K=%d M=%d %dx%d mode %3d: %8.1f us  %7.1f TF/s' % (K, M, H, W, mode, ms.value * 1e3, 2.0 * 9 * K * M * H * W / ms.value / 1e9), flush=True)
