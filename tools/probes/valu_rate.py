#!/usr/bin/env python3
"""Issue cost of the VALU instructions of the split-operand transform, alone and between bf16 MFMAs (one wave per SIMD)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import probes
lib = probes.load_library()
names = ['v_add_f32', 'v_pk_add_f32', 'v_cvt_pk_bf16_f32', 'v_lshlrev_b32', 'v_and_b32', 'v_fma_f32', 'v_perm_b32', 'v_pk_fma_f32', 'v_pk_mul_f32',
         'v_dot2c_f32_bf16', 'ds_read_b32', 'ds_read_b64', 'ds_read2_b32', 'v_sub_f32', 'v_bfe_u32']
v = ctypes.c_double()
for kind, name in enumerate(names):
    row = []
    for nv, mf in ((4, 0), (8, 0), (0, 1), (2, 1), (4, 1), (6, 1), (8, 1)):
        rc = lib.st_probe_valu_rate(0, kind, nv, mf, ctypes.byref(v))
        row.append('%s%d: %6.1f' % ('mfma+' if mf else 'alone ', nv, v.value if rc == 0 else -1))
    print('%-20s %s' % (name, '   '.join(row)), flush=True)
