#!/usr/bin/env python3
"""Cycles per k-pair of the LDS-staged-U Winograd feed pattern (GPU box); 1024 = MFMA-bound."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import probes
lib = probes.load_library()
for K in (512, 128):
    for extra in (0, 1):
        for blocks in (8, 256, 1024):
            c = ctypes.c_double()
            rc = lib.st_bench_lds_feed_probe(0, extra, K, blocks, ctypes.byref(c))
            print('K=%d extra DMA pieces per wave per k-pair=%d blocks=%4d: rc=%d %.0f cycles per k-pair (%.0f %% of the MFMA rate)' % (K, extra, blocks, rc, c.value, 102400.0 / max(c.value, 1)), flush=True)
