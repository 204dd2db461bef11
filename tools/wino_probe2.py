#!/usr/bin/env python3
"""Is the U stream bound per CU or by contention in the XCD's L2?  Same probe with 8, 32, 64, 128, 256 workgroups."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import probes
lib = probes.load_library()
for K, M in ((512, 512), (128, 128)):
    for blocks in (8, 32, 64, 128, 256, 512):
        tf = ctypes.c_double()
        rc = lib.st_bench_wino_probe(0, -blocks, K, M, 2, ctypes.byref(tf))
        cus = min(blocks, 256)
        print('K=%d M=%d blocks=%3d: rc=%d %.1f TF/s executed = %.3f TF/s per busy CU (peak 0.614)' % (K, M, blocks, rc, tf.value, tf.value / cus * (blocks / cus if blocks > 256 else 1) if False else tf.value / cus), flush=True)
