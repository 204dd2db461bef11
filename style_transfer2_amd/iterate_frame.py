"""``messages.Iterate`` as one pickle assembled IN PLACE around the image the GPU copied into pinned memory.

The reference sends every iterate as ``sock_out.send_pyobj(Iterate(image, i, trace))`` (worker.py:351-353,
messages.py:64-74): ``pickle.dumps`` copies the H x W x 3 float32 image twice (``ndarray.tobytes`` and the pickle
buffer) -- 50 MB at 2048 x 2048, the limit of the worker's wire rate there.  A pickle is a program for a stack
machine, and the image bytes are one contiguous operand of it (``BINBYTES <length> <bytes>``), so the frame is

    [ head : PROTO, the Iterate instance, key 'image', the ndarray reconstruct call up to BINBYTES <length> ]
    [ the image, written by the GPU's copy engine (st_step_begin) -- never touched by the host               ]
    [ tail : end of the ndarray state, keys 'i' and 'trace' with their values, SETITEMS, BUILD, STOP          ]

and head / tail are a few hundred bytes the host writes into the room ``st_step_frame_room`` reserves around the
image.  ``pickle.loads(frame)`` yields exactly what ``pickle.loads(pickle.dumps(Iterate(image, i, trace)))`` yields
(same class path, attribute names, array dtype / shape / order, ``i`` an int, ``trace`` an OrderedDict of python
scalars): tests/test_iterate_frame.py.  The bytes differ from ``pickle.dumps``'s (no FRAME opcodes, another memo
numbering); the receiver's ``recv_pyobj`` is ``pickle.loads`` and accepts any valid stream.
"""

import pickle
import struct

import numpy as np

HEAD_ROOM = 4096        # st_step_frame_room: what the worker asks for (head is ~250 bytes, the tail grows with the trace)
TAIL_ROOM = 64 * 1024

_MARK, _TUPLE, _BUILD, _REDUCE, _SETITEMS, _STOP = b'(', b't', b'b', b'R', b'u', b'.'
_EMPTY_TUPLE, _EMPTY_DICT, _NEWOBJ = b')', b'}', b'\x81'


def _push(obj):
    """Opcodes that leave exactly ``obj`` on the unpickler's stack: a complete protocol-3 pickle without PROTO and STOP.
    (Memo slots written by one fragment are only read back inside the same fragment, after it rewrote them.)"""
    data = pickle.dumps(obj, protocol=3)
    assert data[:2] == b'\x80\x03' and data[-1:] == _STOP
    return data[2:-1]


def _bytes_op(n):
    return (b'B' + struct.pack('<I', n)) if n < (1 << 32) else (b'\x8e' + struct.pack('<Q', n))


def head(shape, dtype=np.float32):
    """Everything in front of the image bytes."""
    import messages
    dtype = np.dtype(dtype)
    recon, args, state = np.empty((0,), dtype).__reduce__()       # (_reconstruct, (ndarray, (0,), b'b'), (version, shape, dtype, fortran, data))
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    proto = b'\x80\x03' if n < (1 << 32) else b'\x80\x04'        # BINBYTES8 is a protocol-4 opcode
    return b''.join((
        proto,
        _push(messages.Iterate), _EMPTY_TUPLE, _NEWOBJ,           # Iterate.__new__(Iterate)
        _EMPTY_DICT, _MARK,                                       # its __dict__, filled by SETITEMS below
        _push('image'),
        _push(recon), _push(args), _REDUCE,                       # ndarray shell
        _MARK, _push(state[0]), _push(tuple(int(v) for v in shape)), _push(state[2]), _push(False),
        _bytes_op(n)))                                            # ... the image bytes follow


def tail(i, trace):
    """Everything behind the image bytes."""
    return b''.join((
        _TUPLE, _BUILD,                                           # ndarray.__setstate__((version, shape, dtype, False, bytes))
        _push('i'), _push(int(i)),
        _push('trace'), _push(trace),
        _SETITEMS, _BUILD, _STOP))                                # obj.__dict__.update({...})


def assemble(room, head_room, image_bytes, shape, i, trace):
    """``room``: a writable buffer laid out [head_room | image | tail room] (Engine.step_end(room=True)).  Writes the head
    right-aligned in front of the image and the tail behind it; returns the memoryview of the finished pickle."""
    mv = memoryview(room).cast('B')
    h, t = head(shape), tail(i, trace)
    if len(h) > head_room or head_room + image_bytes + len(t) > len(mv):
        raise ValueError('frame room too small: head %d of %d, tail %d of %d' % (len(h), head_room, len(t), len(mv) - head_room - image_bytes))
    start = head_room - len(h)
    mv[start:head_room] = h
    end = head_room + image_bytes + len(t)
    mv[head_room + image_bytes:end] = t
    return mv[start:end]
