#!/usr/bin/env python3
"""Time one conv shape/config with a chosen iteration count: conv_one.py K M H W cfg dgrad iters [repeat]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import probes
lib = probes.load_library()
K, M, H, W, cfg, dgrad, iters = [int(v) for v in sys.argv[1:8]]
rep = int(sys.argv[8]) if len(sys.argv) > 8 else 1
for _ in range(rep):
    ms, used = ctypes.c_double(), ctypes.c_int()
    rc = lib.st_bench_conv(0, K, M, H, W, cfg, dgrad, iters, ctypes.byref(ms), ctypes.byref(used))
    print('rc=%d cfg=%d %s: %.3f ms  %.1f TF/s' % (rc, used.value, lib.st_conv_config_name(used.value).decode(), ms.value,
                                                   2.0 * 9 * K * M * H * W / ms.value / 1e9), flush=True)
