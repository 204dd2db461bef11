#!/usr/bin/env python3
"""The split-operand Winograd product kernel against the fp32 Winograd kernel, layer by layer (VGG19 shapes at 1024^2 by default):
wino_split_layers.py [edge] -- forward and data-gradient launches, stamps of the main loop."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import probes
lib = probes.load_library()
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def split(K, M, H, W, iters, mode, check=0):
    v = [ctypes.c_double() for _ in range(6)]
    rc = lib.st_probe_wino_split_product(0, K, M, H, W, iters, mode, check, *[ctypes.byref(x) for x in v])
    if rc:
        print('rc=%d %s' % (rc, lib.st_probe_wino_split_error().decode()), flush=True)
        return None
    return [x.value for x in v]


for shape in ((16, 64, 8, 32), (32, 64, 16, 64), (64, 128, 24, 96), (32, 64, 12, 16), (48, 128, 7, 20)):
    r = split(*shape, 2, 0, 1)
    print('check K=%d M=%d %dx%d: rel_l2=%.3g' % (shape + (r[1],)), flush=True)
for shape in ((64, 64, 8, 32), (64, 128, 24, 96), (128, 64, 12, 16)):
    for mode in (4, 5, 6, 7):
        r = split(*shape, 2, mode, 1)
        print('check data gradient (mask %d, inject %d) K=%d M=%d %dx%d: rel_l2=%.3g' % ((mode & 1, (mode >> 1) & 1) + shape + (r[1],)), flush=True)
if len(sys.argv) > 2:
    sys.exit(0)
layers = [('conv1_2', 64, 64, 1), ('conv2_1', 64, 128, 2), ('conv2_2', 128, 128, 2), ('conv3_1', 128, 256, 4), ('conv3_2', 256, 256, 4),
          ('conv4_1', 256, 512, 8), ('conv4_2', 512, 512, 8), ('conv5_1', 512, 512, 16)]
tot = [0.0, 0.0]
for name, cin, cout, div in layers:
    e = edge // div
    for mode, K, M in ((0, cin, cout), (1, cout, cin)):
        r = split(K, M, e, e, 20, mode)
        ms, used = ctypes.c_double(), ctypes.c_int()
        lib.st_bench_conv(0, K, M, e, e, 100, 1 if mode else 0, 20, ctypes.byref(ms), ctypes.byref(used))
        print('%s %s K=%3d M=%3d %4d^2: split %.4f ms (loop %6.0f cycles at %4.0f MHz, prologue %5.0f, epilogue %5.0f)   fp32 %.4f ms   %.2fx'
              % (name, 'dgrad' if mode else 'fwd  ', K, M, e, r[0], r[2], r[3], r[4], r[5], ms.value, ms.value / r[0]), flush=True)
        tot[0] += r[0]; tot[1] += ms.value
print('sum over these 16 launches: split %.3f ms, fp32 %.3f ms: %.2fx' % (tot[0], tot[1], tot[1] / tot[0]))
