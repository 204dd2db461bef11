"""ctypes view of oracle/vgg_layers_ref.c (direct-loop Caffe layer arithmetic).  TEST INFRASTRUCTURE.

An implementation independent of the im2col+SGEMM one in oracle/caffe_net.py; the two are
cross-checked in tests/test_oracle_net.py."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'libvgg_layers_ref.so')
_lib = None
F32 = np.float32


def build(force=False):
    src = os.path.join(_HERE, 'vgg_layers_ref.c')
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.ref_pooled_size.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def conv3x3_forward(x, w, b, relu=False):
    cin, h, wd = x.shape
    cout = w.shape[0]
    x, w, b = (np.ascontiguousarray(a, F32) for a in (x, w, b))
    y = np.empty((cout, h, wd), F32)
    lib().ref_conv3x3_forward(_p(x), _p(w), _p(b), _p(y), cin, cout, h, wd, int(relu))
    return y


def conv3x3_backward_data(dy, w):
    cout, h, wd = dy.shape
    cin = w.shape[1]
    dy, w = np.ascontiguousarray(dy, F32), np.ascontiguousarray(w, F32)
    dx = np.empty((cin, h, wd), F32)
    lib().ref_conv3x3_backward_data(_p(dy), _p(w), _p(dx), cin, cout, h, wd)
    return dx


def maxpool_forward(x):
    c, h, wd = x.shape
    x = np.ascontiguousarray(x, F32)
    ho, wo = lib().ref_pooled_size(h), lib().ref_pooled_size(wd)
    y = np.empty((c, ho, wo), F32)
    arg = np.empty((c, ho, wo), np.int32)
    lib().ref_maxpool_forward(_p(x), _p(y), _p(arg), c, h, wd)
    return y, arg


def maxpool_backward(dy, arg, in_shape):
    c, h, wd = in_shape
    dy = np.ascontiguousarray(dy, F32)
    dx = np.empty((c, h, wd), F32)
    lib().ref_maxpool_backward(_p(dy), _p(arg), _p(dx), c, h, wd)
    return dx
