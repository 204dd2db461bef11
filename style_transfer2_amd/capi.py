"""ctypes binding of include/st2.h (the C ABI of libst2_hip.so)."""

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))


class HipUnavailable(RuntimeError):
    """The HIP library is not built or cannot be loaded.  There is deliberately no fallback."""


class StError(RuntimeError):
    """A non-zero status from the C ABI; the message is st_last_error()."""


class ResampleTable(ctypes.Structure):
    _fields_ = [('lo', POINTER(c_int)), ('n', POINTER(c_int)), ('k', POINTER(c_double)), ('kmax', c_int), ('out_size', c_int)]


class TilePeer(ctypes.Structure):
    _fields_ = [('peer', c_int), ('n_send', c_int), ('send_rects', POINTER(c_int)), ('n_recv', c_int), ('recv_rects', POINTER(c_int))]


# transport callbacks of st_comm_callbacks (device pointers arrive as integers)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int)
EXCHANGE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, POINTER(c_int), POINTER(c_void_p), POINTER(c_int),
                               c_int, POINTER(c_int), POINTER(c_void_p), POINTER(c_int))
COMM_ID_BYTES = 128


class LayerDesc(ctypes.Structure):
    _fields_ = [('kind', c_int), ('name', c_char_p), ('cin', c_int), ('cout', c_int)]


def lib_path():
    return os.environ.get('ST2_HIP_LIB', os.path.join(HERE, 'lib', 'libst2_hip.so'))


# name -> (restype, argtypes); every symbol include/st2.h declares
PROTOTYPES = {
    'st_last_error': (c_char_p, []),
    'st_create': (c_int, [POINTER(c_void_p), c_int, POINTER(LayerDesc), c_int]),
    'st_destroy': (c_int, [c_void_p]),
    'st_load_conv_weights': (c_int, [c_void_p, c_char_p, c_void_p, c_void_p]),
    'st_set_conv_algo': (c_int, [c_void_p, c_int]),
    'st_set_precision': (c_int, [c_void_p, c_int]),
    'st_num_blobs': (c_int, [c_void_p]),
    'st_blob_name': (c_char_p, [c_void_p, c_int]),
    'st_blob_shape': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'st_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int]),
    'st_get_blob': (c_int, [c_void_p, c_int, c_void_p]),
    'st_backward': (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_void_p), c_void_p]),
    'st_gram': (c_int, [c_void_p, c_int, c_void_p]),
    'st_set_input': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int]),
    'st_set_content': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int]),
    'st_set_style': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int]),
    'st_set_input_nchw': (c_int, [c_void_p, c_void_p, c_int, c_int]),
    'st_set_content_nchw': (c_int, [c_void_p, c_void_p, c_int, c_int]),
    'st_get_input_nchw': (c_int, [c_void_p, c_void_p]),
    'st_input_shape': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    'st_set_weights': (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_float), POINTER(c_float),
                               POINTER(c_float), POINTER(c_double)]),
    'st_clear_norms': (c_int, [c_void_p]),
    'st_trace_len': (c_int, [c_void_p]),
    'st_opfunc': (c_int, [c_void_p, POINTER(c_float), c_void_p, c_void_p]),
    'st_optimizer_reset': (c_int, [c_void_p, c_int, c_double]),
    'st_optimizer_set_step': (c_int, [c_void_p, c_double]),
    'st_optimizer_kind': (c_int, [c_void_p]),
    'st_objective_changed': (c_int, [c_void_p]),
    'st_adam_get_state': (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int), POINTER(c_int)]),
    'st_adam_set_state': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    'st_step': (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_float)]),
    'st_step_begin': (c_int, [c_void_p]),
    'st_step_end': (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int), POINTER(c_int), c_void_p, POINTER(c_float)]),
    'st_step_pending': (c_int, [c_void_p]),
    'st_step_frame_room': (c_int, [c_void_p, c_size_t, c_size_t]),
    'st_graph_replays': (c_int, [c_void_p, POINTER(c_longlong)]),
    'st_lbfgs_inv_hv': (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p]),
    'st_sync': (c_int, [c_void_p]),
    'st_profile_enable': (c_int, [c_void_p, c_int]),
    'st_profile_num_classes': (c_int, []),
    'st_profile_class_name': (c_char_p, [c_int]),
    'st_profile_read': (c_int, [c_void_p, POINTER(c_longlong), POINTER(c_double), POINTER(c_double),
                                POINTER(c_double)]),
    'st_resample_state': (c_int, [c_void_p, POINTER(ResampleTable), POINTER(ResampleTable), POINTER(ResampleTable), POINTER(ResampleTable), c_void_p]),
    'st_resample_content': (c_int, [c_void_p, POINTER(ResampleTable), POINTER(ResampleTable)]),
    'st_get_content_nchw': (c_int, [c_void_p, c_void_p, POINTER(c_int), POINTER(c_int)]),
    'st_tile_configure': (c_int, [c_void_p] + [c_int] * 8),
    'st_tile_forward': (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int)]),
    'st_tile_losses': (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int)]),
    'st_tile_style_raw': (c_int, [c_void_p]),
    'st_tile_losses_finish': (c_int, [c_void_p]),
    'st_tile_backward': (c_int, [c_void_p, POINTER(c_void_p)]),
    'st_tile_update': (c_int, [c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_int)]),
    'st_tile_buffer': (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    'st_tile_swap': (c_int, [c_void_p]),
    'st_tile_gradient': (c_int, [c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_int)]),
    'st_vec_dot': (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    'st_vec_axpy': (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_longlong]),
    'st_vec_div': (c_int, [c_void_p, c_double, c_void_p, c_longlong]),
    'st_tile_strips': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int), c_void_p, c_int]),
    'st_comm_unique_id': (c_int, [c_char_p]),
    'st_comm_init': (c_int, [c_void_p, c_char_p, c_int, c_int]),
    'st_comm_callbacks': (c_int, [c_void_p, c_int, c_int, ALLREDUCE_FN, EXCHANGE_FN, c_void_p]),
    'st_comm_destroy': (c_int, [c_void_p]),
    'st_comm_barrier': (c_int, [c_void_p]),
    'st_tile_plan': (c_int, [c_void_p, c_int, c_int, POINTER(TilePeer)]),
    'st_tile_step': (c_int, [c_void_p, c_void_p]),
    'st_tile_get_tile': (c_int, [c_void_p, c_void_p]),
}

_lib = None


def load_library():
    """dlopen libst2_hip.so and attach prototypes.  Raises HipUnavailable when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise HipUnavailable('%s is not built (run `python -c "import __graft_entry__ as g; g.build()"` '
                             'or `python style_transfer2_amd/build.py`)' % path)
    try:
        lib = ctypes.CDLL(path)
    except OSError as err:
        raise HipUnavailable('cannot load %s: %s' % (path, err))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)        # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load_library().st_last_error()
        raise StError('st2 error %d: %s' % (status, msg.decode() if msg else '?'))
