#!/usr/bin/env python3
"""Matrix-pipe ceiling on this chip for v_mfma_f32_32x32x2_f32 (GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import probes
lib = probes.load_library()
for variant, name in ((0, 'registers, smooth data'), (4, 'registers, random data'), (1, 'LDS reads, smooth data'), (2, 'LDS reads, random data'), (3, 'LDS reads, random, pinned order')):
    for bpc in (1, 2, 3):
        tf = ctypes.c_double()
        rc = lib.st_bench_mfma(0, variant, bpc, ctypes.byref(tf))
        print('%-34s %d blocks/CU: %6.1f TFLOP/s' % (name, bpc, tf.value), flush=True)
