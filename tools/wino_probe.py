#!/usr/bin/env python3
"""Can transformed Winograd weights stream from L2 straight into MFMA operand registers fast enough? (GPU box)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import probes
lib = probes.load_library()
for K, M in ((512, 512), (256, 256), (128, 128)):
    for depth in (1, 2, 12):
        for bpc in (2, 4):
            tf = ctypes.c_double()
            rc = lib.st_bench_wino_probe(0, bpc, K, M, depth, ctypes.byref(tf))
            print('K=%d M=%d depth=%d blocks/CU=%d: rc=%d executed %.1f TFLOP/s (x2.25 = %.1f effective)' % (K, M, depth, bpc, rc, tf.value, tf.value * 2.25), flush=True)
