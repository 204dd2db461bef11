"""Thin object view of the C ABI: one ``Engine`` == one ``st_ctx`` == one reference worker's
model + objective + optimizer, resident on one MI355X."""

import ctypes
from ctypes import POINTER, byref, c_double, c_float, c_int, c_longlong, c_void_p

import numpy as np

from . import capi
from .capi import check

F32 = np.float32
OPT_ADAM, OPT_LBFGS = 1, 2

# reference models/vgg19.prototxt:1-337
VGG19_TOPOLOGY = (
    ('conv', 'conv1_1', 3, 64), ('conv', 'conv1_2', 64, 64), ('pool', 'pool1'),
    ('conv', 'conv2_1', 64, 128), ('conv', 'conv2_2', 128, 128), ('pool', 'pool2'),
    ('conv', 'conv3_1', 128, 256), ('conv', 'conv3_2', 256, 256),
    ('conv', 'conv3_3', 256, 256), ('conv', 'conv3_4', 256, 256), ('pool', 'pool3'),
    ('conv', 'conv4_1', 256, 512), ('conv', 'conv4_2', 512, 512),
    ('conv', 'conv4_3', 512, 512), ('conv', 'conv4_4', 512, 512), ('pool', 'pool4'),
    ('conv', 'conv5_1', 512, 512), ('conv', 'conv5_2', 512, 512),
    ('conv', 'conv5_3', 512, 512), ('conv', 'conv5_4', 512, 512), ('pool', 'pool5'),
)


def _ptr(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


def _image_arg(image):
    """HxWx3 uint8 or anything float-convertible -> (contiguous array, is_u8)."""
    arr = np.asarray(image)
    if arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError('image must be HxWx3, got %s' % (arr.shape,))
    if arr.dtype == np.uint8:
        return np.ascontiguousarray(arr), 1
    return np.ascontiguousarray(arr, F32), 0


class Engine:
    def __init__(self, topology=None, device=0, precision='fp32'):
        self.lib = capi.load_library()
        self.precision = precision
        self._ctx = c_void_p()
        self.topology = tuple(topology) if topology is not None else VGG19_TOPOLOGY
        if topology is None:
            check(self.lib.st_create(byref(self._ctx), int(device), None, 0))
        else:
            descs = (capi.LayerDesc * len(self.topology))()
            self._names = []          # keep the byte strings alive
            for d, layer in zip(descs, self.topology):
                name = layer[1].encode()
                self._names.append(name)
                d.kind = 0 if layer[0] == 'conv' else 1
                d.name = name
                d.cin, d.cout = (layer[2], layer[3]) if layer[0] == 'conv' else (0, 0)
            check(self.lib.st_create(byref(self._ctx), int(device), descs, len(self.topology)))
        # 'bf16-full' = the bf16 feature path with every fp32 blob / diff materialised (A/B reference of the lean data flow)
        self.set_precision(precision)
        n = self.lib.st_num_blobs(self._ctx)
        self.blob_names = [self.lib.st_blob_name(self._ctx, i).decode() for i in range(n)]
        self._index = {name: i for i, name in enumerate(self.blob_names)}

    PRECISIONS = {'fp32': 0, 'bf16': 1, 'bf16-full': 2}

    def set_precision(self, precision):
        """'fp32', 'bf16' (lean data flow: blobs only bf16 convs read are not materialised in fp32) or 'bf16-full' (same arithmetic,
        every blob / diff written in fp32 as well).  Takes effect from the next evaluation."""
        if precision not in self.PRECISIONS:
            raise ValueError('precision must be one of %s' % sorted(self.PRECISIONS))
        check(self.lib.st_set_precision(self._ctx, self.PRECISIONS[precision]))
        self.precision = precision

    def graph_replays(self):
        """Number of steps that ran as a hipGraph replay (steady-state Adam steps at small image sizes)."""
        n = c_longlong()
        check(self.lib.st_graph_replays(self._ctx, byref(n)))
        return n.value

    def set_conv_algo(self, winograd):
        """True / 1 (default): eligible fp32 convs run as Winograd F(2x2,3x3) on the fp32 matrix cores; False / 0: direct kernel only;
        2: Winograd with the transform-domain products as six bf16 partial products of split operands (fp32 results) where the shape allows."""
        check(self.lib.st_set_conv_algo(self._ctx, 2 if winograd == 2 else (1 if winograd else 0)))

    # -- lifecycle -------------------------------------------------------------------------------
    def close(self):
        if self._ctx:
            self.lib.st_destroy(self._ctx)
            self._ctx = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model -----------------------------------------------------------------------------------
    def load_weights(self, params):
        """params: {conv name: (w (Cout,Cin,3,3), b (Cout,))}"""
        for layer in self.topology:
            if layer[0] != 'conv':
                continue
            w, b = params[layer[1]]
            w = np.ascontiguousarray(w, F32)
            b = np.ascontiguousarray(b, F32)
            if w.shape != (layer[3], layer[2], 3, 3) or b.shape != (layer[3],):
                raise ValueError('%s: bad weight shapes %s %s' % (layer[1], w.shape, b.shape))
            check(self.lib.st_load_conv_weights(self._ctx, layer[1].encode(), _ptr(w), _ptr(b)))

    def blob_index(self, name):
        return self._index[name]

    def blob_shape(self, name, h, w):
        c, bh, bw = c_int(), c_int(), c_int()
        check(self.lib.st_blob_shape(self._ctx, self._index[name], h, w, byref(c), byref(bh), byref(bw)))
        return c.value, bh.value, bw.value

    def forward(self, x_nchw, last=None):
        x = np.ascontiguousarray(x_nchw, F32)
        assert x.ndim == 4 and x.shape[:2] == (1, 3)
        self._fwd_hw = x.shape[2:]
        last_i = -1 if last is None else self._index[last]
        check(self.lib.st_forward(self._ctx, _ptr(x), x.shape[2], x.shape[3], last_i))

    def get_blob(self, name):
        """Host copy of a blob of the last forward (st_forward, or the forward inside the last opfunc / step)."""
        h, w = getattr(self, '_fwd_hw', None) or self.input_shape()
        out = np.empty((1,) + self.blob_shape(name, h, w), F32)
        check(self.lib.st_get_blob(self._ctx, self._index[name], _ptr(out)))
        return out

    def backward(self, diffs):
        h, w = self._fwd_hw
        names = list(diffs)
        arrs = [np.ascontiguousarray(diffs[n], F32) for n in names]
        for n, a in zip(names, arrs):
            if a.shape != (1,) + self.blob_shape(n, h, w):
                raise ValueError('diff for %s has shape %s' % (n, a.shape))
        idx = (c_int * len(names))(*[self._index[n] for n in names])
        ptrs = (c_void_p * len(names))(*[a.ctypes.data for a in arrs])
        out = np.empty((1, 3, h, w), F32)
        check(self.lib.st_backward(self._ctx, len(names), idx, ptrs, _ptr(out)))
        return out

    def gram(self, name):
        c = self.blob_shape(name, *self._fwd_hw)[0]
        out = np.empty((c, c), F32)
        check(self.lib.st_gram(self._ctx, self._index[name], _ptr(out)))
        return out

    # -- image slots -------------------------------------------------------------------------------
    def set_input(self, image):
        arr, u8 = _image_arg(image)
        check(self.lib.st_set_input(self._ctx, _ptr(arr), arr.shape[0], arr.shape[1], u8))

    def set_content(self, image):
        arr, u8 = _image_arg(image)
        check(self.lib.st_set_content(self._ctx, _ptr(arr), arr.shape[0], arr.shape[1], u8))

    def set_style(self, image):
        arr, u8 = _image_arg(image)
        check(self.lib.st_set_style(self._ctx, _ptr(arr), arr.shape[0], arr.shape[1], u8))

    def set_input_nchw(self, x):
        x = np.ascontiguousarray(x, F32)
        check(self.lib.st_set_input_nchw(self._ctx, _ptr(x), x.shape[2], x.shape[3]))

    def set_content_nchw(self, x):
        x = np.ascontiguousarray(x, F32)
        check(self.lib.st_set_content_nchw(self._ctx, _ptr(x), x.shape[2], x.shape[3]))

    def input_shape(self):
        h, w = c_int(), c_int()
        check(self.lib.st_input_shape(self._ctx, byref(h), byref(w)))
        return h.value, w.value

    def get_input_nchw(self):
        h, w = self.input_shape()
        out = np.empty((1, 3, h, w), F32)
        check(self.lib.st_get_input_nchw(self._ctx, _ptr(out)))
        return out

    # -- objective -----------------------------------------------------------------------------------
    def set_weights(self, rows, content, style, deepdream, params4):
        n = len(rows)
        idx = (c_int * n)(*[self._index[r] for r in rows])
        cw = (c_float * n)(*[float(v) for v in content])
        sw = (c_float * n)(*[float(v) for v in style])
        dw = (c_float * n)(*[float(v) for v in deepdream])
        pr = (c_double * 4)(*[float(v) for v in params4])
        check(self.lib.st_set_weights(self._ctx, n, idx, cw, sw, dw, pr))

    def clear_norms(self):
        check(self.lib.st_clear_norms(self._ctx))

    def trace_len(self):
        return self.lib.st_trace_len(self._ctx)

    def opfunc(self, return_grad=True):
        h, w = self.input_shape()
        loss = c_float()
        trace = np.zeros(self.trace_len(), np.float64)
        grad = np.empty((1, 3, h, w), F32) if return_grad else None
        check(self.lib.st_opfunc(self._ctx, byref(loss), _ptr(grad), _ptr(trace)))
        self._fwd_hw = (h, w)           # the blobs up to the deepest weighted layer are those of this evaluation
        return F32(loss.value), grad, trace

    # -- optimizer -----------------------------------------------------------------------------------
    def optimizer_reset(self, kind, step_size):
        check(self.lib.st_optimizer_reset(self._ctx, kind, float(step_size)))

    def optimizer_set_step(self, step_size):
        check(self.lib.st_optimizer_set_step(self._ctx, float(step_size)))

    def objective_changed(self):
        check(self.lib.st_objective_changed(self._ctx))

    def adam_get_state(self):
        h, w = self.input_shape()
        m = np.empty((1, 3, h, w), F32)
        v = np.empty((1, 3, h, w), F32)
        i1, i2 = c_int(), c_int()
        check(self.lib.st_adam_get_state(self._ctx, _ptr(m), _ptr(v), byref(i1), byref(i2)))
        return m, v, i1.value, i2.value

    def adam_set_state(self, m, v, items1, items2):
        m = np.ascontiguousarray(m, F32)
        v = np.ascontiguousarray(v, F32)
        check(self.lib.st_adam_set_state(self._ctx, _ptr(m), _ptr(v), int(items1), int(items2)))

    # -- device-resident resampling (Pillow-exact) -------------------------------------------------------
    @staticmethod
    def _table(in_size, out_size, method):
        from . import resample
        lo, n, k = resample.pillow_coeffs(in_size, out_size, method)
        t = capi.ResampleTable()
        t._keep = (np.ascontiguousarray(lo), np.ascontiguousarray(n), np.ascontiguousarray(k))
        t.lo = t._keep[0].ctypes.data_as(ctypes.POINTER(c_int))
        t.n = t._keep[1].ctypes.data_as(ctypes.POINTER(c_int))
        t.k = t._keep[2].ctypes.data_as(ctypes.POINTER(c_double))
        t.kmax, t.out_size = k.shape[1], out_size
        return t

    def resample_state(self, size, new_x=None):
        """optimizer.resample (optimizers.py:29-40,110-119) on the device: x (Lanczos, or replaced by new_x),
        Adam m (Lanczos), Adam v (bilinear, clipped at 0)."""
        from . import resample
        h, w = self.input_shape()
        if new_x is not None:
            new_x = np.ascontiguousarray(new_x, F32)
            size = new_x.shape[2:]
        size = tuple(int(v) for v in size)
        lx, ly = self._table(w, size[1], resample.LANCZOS), self._table(h, size[0], resample.LANCZOS)
        bx, by = self._table(w, size[1], resample.BILINEAR), self._table(h, size[0], resample.BILINEAR)
        check(self.lib.st_resample_state(self._ctx, byref(lx), byref(ly), byref(bx), byref(by), _ptr(new_x)))

    def resample_content(self, size):
        from . import resample
        h, w = self.content_shape()
        size = tuple(int(v) for v in size)
        lx, ly = self._table(w, size[1], resample.LANCZOS), self._table(h, size[0], resample.LANCZOS)
        check(self.lib.st_resample_content(self._ctx, byref(lx), byref(ly)))

    def content_shape(self):
        h, w = c_int(), c_int()
        check(self.lib.st_get_content_nchw(self._ctx, None, byref(h), byref(w)))
        return h.value, w.value

    def get_content_nchw(self):
        h, w = self.content_shape()
        out = np.empty((1, 3, h, w), F32)
        check(self.lib.st_get_content_nchw(self._ctx, _ptr(out), None, None))
        return out

    def step(self, want_image=True, want_trace=True):
        """One iteration.  With both flags False nothing is read back and the call is asynchronous."""
        h, w = self.input_shape()
        img = np.empty((h, w, 3), F32) if want_image else None
        trace = np.zeros(self.trace_len(), np.float64) if want_trace else None
        loss = c_float()
        check(self.lib.st_step(self._ctx, _ptr(img), _ptr(trace),
                               byref(loss) if (want_image or want_trace) else None))
        return img, trace, F32(loss.value)

    def step_begin(self):
        """Queue one iteration and the copy of its iterate / trace to the host (st_step_begin); at most two may be in flight."""
        check(self.lib.st_step_begin(self._ctx))

    STEP_VIEW_LIFETIME = 5      # st2.h: a view returned by step_end(copy=False) survives this many further step_begin calls

    def set_frame_room(self, head_bytes, tail_bytes):
        """Reserve room in front of / behind the iterates step_end hands out (st_step_frame_room), from the next step_begin on."""
        check(self.lib.st_step_frame_room(self._ctx, int(head_bytes), int(tail_bytes)))
        self._frame_room = ((int(head_bytes) + 4095) // 4096 * 4096, int(tail_bytes))

    def step_end(self, copy=True, room=False):
        """Results of the OLDEST queued iteration, as step() returns them.  copy=False returns a read-only view of the context's
        pinned buffer instead of an owned array: valid for the next STEP_VIEW_LIFETIME calls of step_begin (the worker's bounded
        sender queue stays inside that; a 12.6 MB host copy per iterate is what it saves).  room=True (with copy=False, after
        set_frame_room): a fourth result, the writable buffer [head room | image | tail room] the view lies in (iterate_frame.py)."""
        ptr, h, w, loss = c_void_p(), c_int(), c_int(), c_float()
        trace = np.zeros(self.trace_len(), np.float64)
        check(self.lib.st_step_end(self._ctx, byref(ptr), byref(h), byref(w), _ptr(trace), byref(loss)))
        view = np.ctypeslib.as_array(ctypes.cast(ptr, POINTER(c_float)), shape=(h.value, w.value, 3))
        if copy:
            return view.copy(), trace, F32(loss.value)
        view.flags.writeable = False
        if room:
            head, tail = getattr(self, '_frame_room', (0, 0))
            buf = (ctypes.c_char * (head + view.nbytes + tail)).from_address(ptr.value - head)
            return view, trace, F32(loss.value), buf
        return view, trace, F32(loss.value)

    def steps_pending(self):
        return int(self.lib.st_step_pending(self._ctx))

    def lbfgs_inv_hv(self, pairs, g):
        """Test hook: H g of optimizers.py:89-108 for the given [(s, y), ...] history (oldest first) on the device."""
        g = np.ascontiguousarray(g, F32)
        ss = [np.ascontiguousarray(s, F32) for s, _ in pairs]
        ys = [np.ascontiguousarray(y, F32) for _, y in pairs]
        sp = (c_void_p * len(pairs))(*[a.ctypes.data for a in ss])
        yp = (c_void_p * len(pairs))(*[a.ctypes.data for a in ys])
        out = np.empty_like(g)
        check(self.lib.st_lbfgs_inv_hv(self._ctx, len(pairs), sp, yp, _ptr(g), _ptr(out)))
        return out

    def sync(self):
        check(self.lib.st_sync(self._ctx))

    # -- measurement -----------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        check(self.lib.st_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self):
        n = self.lib.st_profile_num_classes()
        launches = (c_longlong * n)()
        ms, flops, nbytes = (c_double * n)(), (c_double * n)(), (c_double * n)()
        check(self.lib.st_profile_read(self._ctx, launches, ms, flops, nbytes))
        out = {}
        for i in range(n):
            if launches[i]:
                out[self.lib.st_profile_class_name(i).decode()] = dict(
                    launches=int(launches[i]), ms=float(ms[i]), flops=float(flops[i]), bytes=float(nbytes[i]))
        return out
