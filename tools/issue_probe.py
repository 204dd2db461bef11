#!/usr/bin/env python3
"""Shader cycles per fp32 32x32x2 MFMA with one wave per SIMD and auxiliary instructions in between (GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import probes
lib = probes.load_library()
for naux, nlds in ((0, 0), (2, 0), (4, 0), (8, 0), (12, 0), (0, 2), (0, 4), (4, 2), (4, 4)):
    c = ctypes.c_double()
    rc = lib.st_bench_issue_probe(0, naux, nlds, ctypes.byref(c))
    print('VALU %2d  ds_read %d per MFMA: rc=%d  %.1f cycles per MFMA' % (naux, nlds, rc, c.value), flush=True)
