"""ctypes view of tools/probes/libst2_probes.so (tools/probes/st2_probes.h): development probes, not product code."""
import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_double, c_int

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

PROTOTYPES = {
    'st_probe_last_error': (c_char_p, []),
    'st_bench_conv': (c_int, [c_int] * 8 + [POINTER(c_double), POINTER(c_int)]),
    'st_conv_num_configs': (c_int, []),
    'st_conv_config_name': (c_char_p, [c_int]),
    'st_bench_mfma': (c_int, [c_int, c_int, c_int, POINTER(c_double)]),
    'st_bench_issue_probe': (c_int, [c_int, c_int, c_int, POINTER(c_double)]),
    'st_bench_lds_feed_probe': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_double)]),
    'st_bench_conv16': (c_int, [c_int] * 7 + [POINTER(c_double)]),
    'st_bench_wino_probe': (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_double)]),
    'st_probe_wino_split': (c_int, [c_int] * 7 + [POINTER(c_double)] * 6),
    'st_probe_wino_split_error': (c_char_p, []),
    'st_probe_wino_split_product': (c_int, [c_int] * 8 + [POINTER(c_double)] * 6),
    'st_probe_valu_rate': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_double)]),
}


def load_library():
    from style_transfer2_amd import build
    lib = ctypes.CDLL(build.build_probes())
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib
