"""Minimal reader (and test writer) for Caffe ``.caffemodel`` files -- protobuf wire format, no Caffe,
no generated code (reference worker.py:58-61 ``caffe.Net(prototxt, 1, weights=)``,
config.ini:28-29, download_models.sh:3).

Only what a weight file needs is understood (field numbers from BVLC caffe.proto):
  NetParameter    : layer = 100 (LayerParameter), layers = 2 (V1LayerParameter)
  LayerParameter  : name = 1, blobs = 7          V1LayerParameter: name = 4, blobs = 6
  BlobProto       : num/channels/height/width = 1..4, data = 5 (float, packed or not),
                    shape = 7 (BlobShape: dim = 1), double_data = 8
Returns ``{layer name: [ndarray, ...]}``; ``vgg_params`` turns that into the engine's
``{conv: (w (Cout,Cin,3,3), b (Cout,))}``.  The real vgg19.caffemodel cannot be fetched offline, so the
RGB channel-order assumption of the reference (worker.py:34,66: RGB means, no flip) stays unverified;
``bgr_to_rgb=True`` flips the first conv's input channels for the stock BGR VGG19 file.
"""

import struct
from collections import OrderedDict

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------- wire format
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _fields(buf):
    """Yields (field number, wire type, value) of one message; value is int or a memoryview slice."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError('unsupported wire type %d' % wt)
        if pos > end:
            raise ValueError('truncated message')
        yield field, wt, val


def _blob(buf):
    legacy = [None] * 4
    shape = None
    chunks, singles, doubles = [], [], []
    for field, wt, val in _fields(buf):
        if 1 <= field <= 4 and wt == 0:
            legacy[field - 1] = val
        elif field == 5:
            if wt == 2:
                chunks.append(np.frombuffer(val, '<f4'))
            else:
                singles.append(struct.unpack('<f', bytes(val))[0])
        elif field == 8:
            doubles.append(np.frombuffer(val, '<f8') if wt == 2 else np.array(struct.unpack('<d', bytes(val))))
        elif field == 7 and wt == 2:
            dims = []
            for f2, w2, v2 in _fields(val):
                if f2 == 1:
                    if w2 == 2:
                        p = 0
                        while p < len(v2):
                            d, p = _varint(v2, p)
                            dims.append(d)
                    else:
                        dims.append(v2)
            shape = dims
    if singles:
        chunks.append(np.asarray(singles, F32))
    data = np.concatenate(chunks).astype(F32) if chunks else (
        np.concatenate(doubles).astype(F32) if doubles else np.zeros(0, F32))
    if shape is None and any(v is not None for v in legacy):
        shape = [1 if v is None else v for v in legacy]
    if shape is not None and int(np.prod(shape)) == data.size:
        data = data.reshape(shape)
    return data


def read_caffemodel(path_or_bytes):
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, 'rb').read()
    buf = memoryview(raw)
    layers = OrderedDict()
    for field, wt, val in _fields(buf):
        if wt != 2 or field not in (100, 2):
            continue
        name_field, blob_field = (1, 7) if field == 100 else (4, 6)
        name, blobs = None, []
        for f2, w2, v2 in _fields(val):
            if f2 == name_field and w2 == 2:
                name = bytes(v2).decode('utf-8')
            elif f2 == blob_field and w2 == 2:
                blobs.append(_blob(v2))
        if name is not None and blobs:
            layers[name] = blobs
    return layers


def vgg_params(layers, topology, bgr_to_rgb=False):
    params = OrderedDict()
    first = True
    for layer in topology:
        if layer[0] != 'conv':
            continue
        _, name, cin, cout = layer
        if name not in layers or len(layers[name]) < 2:
            raise KeyError('caffemodel has no weights for %s' % name)
        w = np.asarray(layers[name][0], F32).reshape(cout, cin, 3, 3)
        b = np.asarray(layers[name][1], F32).reshape(cout)
        if first and bgr_to_rgb:
            w = np.ascontiguousarray(w[:, ::-1])
        first = False
        params[name] = (np.ascontiguousarray(w), b)
    return params


# ------------------------------------------------------------------------------ writer (tests, export)
def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_len(field, payload):
    return _enc_varint(field << 3 | 2) + _enc_varint(len(payload)) + payload


def _enc_blob(arr, legacy_dims=False, packed=True):
    arr = np.ascontiguousarray(arr, '<f4')
    out = b''
    if legacy_dims:
        dims = [1] * (4 - arr.ndim) + list(arr.shape)
        for i, d in enumerate(dims):
            out += _enc_varint((i + 1) << 3) + _enc_varint(d)
    else:
        out += _enc_len(7, _enc_len(1, b''.join(_enc_varint(d) for d in arr.shape)))
    if packed:
        out += _enc_len(5, arr.tobytes())
    else:
        out += b''.join(_enc_varint(5 << 3 | 5) + struct.pack('<f', v) for v in arr.ravel())
    return out


def write_caffemodel(path, params, v1=False, legacy_dims=False, packed=True):
    """params: {layer: (w, b)} -> NetParameter bytes (``layer`` = 100, or V1 ``layers`` = 2)."""
    net = _enc_len(1, b'vgg19_truncated')
    for name, blobs in params.items():
        if v1:
            body = _enc_len(4, name.encode()) + b''.join(_enc_len(6, _enc_blob(b, legacy_dims, packed)) for b in blobs)
            net += _enc_len(2, body)
        else:
            body = _enc_len(1, name.encode()) + _enc_len(2, b'Convolution')
            body += b''.join(_enc_len(7, _enc_blob(b, legacy_dims, packed)) for b in blobs)
            net += _enc_len(100, body)
    if path is not None:
        with open(path, 'wb') as f:
            f.write(net)
    return net
