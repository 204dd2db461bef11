#!/usr/bin/env python3
"""Split-operand Winograd probe against the fp32 Winograd kernel on the same shape:
wino_split.py [K M H W iters] -- first a small checked shape, then the timed one, then conv3x3_wino_f32_128x128 (cfg 101) on it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import probes
lib = probes.load_library()


def split(K, M, H, W, iters, check):
    v = [ctypes.c_double() for _ in range(6)]
    rc = lib.st_probe_wino_split(0, K, M, H, W, iters, check, *[ctypes.byref(x) for x in v])
    if rc:
        print('rc=%d %s' % (rc, lib.st_probe_wino_split_error().decode()), flush=True)
        return None
    ms, err, cyc, mhz, pro, epi = [x.value for x in v]
    print('split  K=%d M=%d %dx%d: %.4f ms  %.1f TF/s algorithmic  rel_l2=%.3g  loop %.0f cycles at %.0f MHz (prologue %.0f, epilogue %.0f)'
          % (K, M, H, W, ms, 2.0 * 9 * K * M * H * W / ms / 1e9, err, cyc, mhz, pro, epi), flush=True)
    return ms


args = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [256, 256, 256, 256, 50]
K, M, H, W, iters = args
for shape in ((16, 64, 8, 32), (32, 64, 16, 64), (64, 128, 24, 96)):
    split(*shape, 2, 1)
t_split = split(K, M, H, W, iters, 0)
for var in (1, 2, 3, 4, 6, 7):
    print('loop experiment %d (1 = no U refills, 2 = no B builds, 4 = no raw reads / row transform):' % var, end=' ')
    split(K, M, H, W, iters, 100 + var)
ms, used = ctypes.c_double(), ctypes.c_int()
for cfg in (101, 100):
    rc = lib.st_bench_conv(0, K, M, H, W, cfg, 0, iters, ctypes.byref(ms), ctypes.byref(used))
    print('fp32   rc=%d cfg=%d %s: %.4f ms  %.1f TF/s algorithmic%s' % (rc, used.value, lib.st_conv_config_name(used.value).decode(), ms.value,
          2.0 * 9 * K * M * H * W / ms.value / 1e9, '   ratio %.2fx' % (ms.value / t_split) if t_split else ''), flush=True)
