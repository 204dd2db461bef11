#!/usr/bin/env python3
"""Writes tests/golden/caffemodel_*.bin: small .caffemodel files encoded by google.protobuf's own serializer from a
hand-written minimal schema (field numbers of BVLC caffe.proto: NetParameter.layers = 2 / .layer = 100,
LayerParameter.name = 1 / .blobs = 7, V1LayerParameter.name = 4 / .blobs = 6, BlobProto 1-5, 7, 8) -- an encoder
independent of style_transfer2_amd/caffemodel.py's own test writer.  Four encodings: V2 layers with BlobShape +
packed floats; V1 layers with legacy num/channels/height/width + unpacked floats; V2 with double_data; V2 with
extra fields the reader must skip (type, bottom, top, phase, an unknown sub-message, a fixed64).
The arrays themselves come from weights.he_normal(TOPO, seed=4, bias_std=0.3) (the test regenerates and compares)."""
import os
import sys

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
TOPO = (('conv', 'conv1_1', 3, 8), ('conv', 'conv1_2', 8, 8), ('pool', 'pool1'), ('conv', 'conv2_1', 8, 16))
T = descriptor_pb2.FieldDescriptorProto


def schema(packed_floats):
    f = descriptor_pb2.FileDescriptorProto(name='mini_caffe_%d.proto' % packed_floats, package='mini%d' % packed_floats, syntax='proto2')

    def msg(name, fields):
        m = f.message_type.add(name=name)
        for fname, num, ftype, label, tname, packed in fields:
            fd = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                fd.type_name = '.mini%d.%s' % (packed_floats, tname)
            if packed is not None:
                fd.options.packed = packed
    opt, rep = T.LABEL_OPTIONAL, T.LABEL_REPEATED
    msg('BlobShape', [('dim', 1, T.TYPE_INT64, rep, None, True)])
    msg('BlobProto', [('num', 1, T.TYPE_INT32, opt, None, None), ('channels', 2, T.TYPE_INT32, opt, None, None),
                      ('height', 3, T.TYPE_INT32, opt, None, None), ('width', 4, T.TYPE_INT32, opt, None, None),
                      ('data', 5, T.TYPE_FLOAT, rep, None, bool(packed_floats)), ('diff', 6, T.TYPE_FLOAT, rep, None, True),
                      ('shape', 7, T.TYPE_MESSAGE, opt, 'BlobShape', None), ('double_data', 8, T.TYPE_DOUBLE, rep, None, True)])
    msg('ParamSpec', [('name', 1, T.TYPE_STRING, opt, None, None), ('lr_mult', 3, T.TYPE_FLOAT, opt, None, None)])
    msg('LayerParameter', [('name', 1, T.TYPE_STRING, opt, None, None), ('type', 2, T.TYPE_STRING, opt, None, None),
                           ('bottom', 3, T.TYPE_STRING, rep, None, None), ('top', 4, T.TYPE_STRING, rep, None, None),
                           ('param', 6, T.TYPE_MESSAGE, rep, 'ParamSpec', None), ('blobs', 7, T.TYPE_MESSAGE, rep, 'BlobProto', None),
                           ('phase', 10, T.TYPE_INT32, opt, None, None), ('debug_tag', 150, T.TYPE_FIXED64, opt, None, None)])
    msg('V1LayerParameter', [('bottom', 2, T.TYPE_STRING, rep, None, None), ('top', 3, T.TYPE_STRING, rep, None, None),
                             ('name', 4, T.TYPE_STRING, opt, None, None), ('type', 5, T.TYPE_INT32, opt, None, None),
                             ('blobs', 6, T.TYPE_MESSAGE, rep, 'BlobProto', None), ('blobs_lr', 7, T.TYPE_FLOAT, rep, None, None)])
    msg('NetParameter', [('name', 1, T.TYPE_STRING, opt, None, None), ('layers', 2, T.TYPE_MESSAGE, rep, 'V1LayerParameter', None),
                         ('input', 3, T.TYPE_STRING, rep, None, None), ('force_backward', 5, T.TYPE_BOOL, opt, None, None),
                         ('layer', 100, T.TYPE_MESSAGE, rep, 'LayerParameter', None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName('mini%d.%s' % (packed_floats, n)))
            for n in ('NetParameter', 'BlobProto')}


def fill_blob(b, arr, how):
    arr = np.ascontiguousarray(arr, np.float32)
    if how == 'legacy':
        dims = [1] * (4 - arr.ndim) + list(arr.shape)
        b.num, b.channels, b.height, b.width = dims
    else:
        b.shape.dim.extend(arr.shape)
    if how == 'double':
        b.double_data.extend(float(v) for v in arr.ravel())
    else:
        b.data.extend(float(v) for v in arr.ravel())


def encode(params, variant):
    cls = schema(packed_floats=0 if variant == 'v1_legacy_unpacked' else 1)
    net = cls['NetParameter'](name='mini_vgg', force_backward=True)
    net.input.append('data')
    prev = 'data'
    for name, (w, b) in params.items():
        if variant == 'v1_legacy_unpacked':
            layer = net.layers.add(name=name, type=4)
            layer.bottom.append(prev); layer.top.append(name); layer.blobs_lr.extend([1.0, 2.0])
            how = 'legacy'
        else:
            layer = net.layer.add(name=name, type='Convolution')
            how = 'double' if variant == 'v2_double' else 'shape'
            if variant == 'v2_extra_fields':
                layer.bottom.append(prev); layer.top.append(name); layer.phase = 1; layer.debug_tag = 0x1122334455667788
                layer.param.add(name=name + '_w', lr_mult=1.0); layer.param.add(name=name + '_b', lr_mult=2.0)
        for arr in (w, b):
            fill_blob(layer.blobs.add(), arr, how)
        if variant == 'v2_extra_fields':          # a weight-less layer in between (ReLU): must not appear in the result
            net.layer.add(name='relu_' + name, type='ReLU').bottom.append(name)
        prev = name
    return net.SerializeToString(deterministic=True)


VARIANTS = ('v2_shape_packed', 'v1_legacy_unpacked', 'v2_double', 'v2_extra_fields')

if __name__ == '__main__':
    from style_transfer2_amd import weights
    params = weights.he_normal(TOPO, seed=4, bias_std=0.3)
    for v in VARIANTS:
        raw = encode(params, v)
        open(os.path.join(HERE, 'caffemodel_%s.bin' % v), 'wb').write(raw)
        print(v, len(raw), 'bytes')
