"""CPU restatement of the network the reference drives through pycaffe.  TEST INFRASTRUCTURE.

Follows (reference file:line)
  * topology            models/vgg19.prototxt:1-337  (16x [Conv3x3 pad1 + in-place ReLU], 5x MaxPool 2x2/2)
  * pre/deprocess       worker.py:34,63-71           (RGB, mean (123.68,116.779,103.939), no channel flip)
  * layer (blob) list   worker.py:73-75
  * forward             worker.py:77-86              (whole net; blobs hold POST-ReLU values)
  * ranged backward     worker.py:88-106             (see ``NetOracle.backward``)

The arithmetic itself lives in BVLC Caffe, which is absent from /root/reference (un-vendored,
un-pinned; located at run time through ``caffe_path``, config.ini:7).  What is restated here is
Caffe's published CPU algorithm: convolution = im2col + SGEMM with K ordered (c, ky, kx),
bottom-diff = W^T * top_diff then col2im; ReLU backward = top_diff * (data > 0); max pooling with
output size ceil((h - 2) / 2) + 1, windows clipped at the border, arg-max = first strictly greater
element in a row-major scan.  Parity at this boundary is UNPINNED by the reference (it has no tests
and Caffe cannot run here); tests/test_oracle_net.py pins it against torch CPU ops and hand cases.
"""

from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
import os

import numpy as np

F32 = np.float32

# Optional callable invoked after every layer of a forward / backward pass (the test session on a GPU box writes a progress
# file from it: minutes-long full-size evaluations then show as progress, and only as long as they do progress).
progress = None


def _tick():
    if progress is not None:
        progress()


# (kind, name, cin, cout) -- models/vgg19.prototxt
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


def tiny_topology(widths=(8, 16), convs_per_stage=(2, 1), final_pool=False):
    """A VGG-shaped miniature (same layer kinds and naming) for second-scale tests."""
    topo, cin = [], 3
    for s, (width, nconv) in enumerate(zip(widths, convs_per_stage), start=1):
        for j in range(1, nconv + 1):
            topo.append(('conv', 'conv%d_%d' % (s, j), cin, width))
            cin = width
        if s < len(widths) or final_pool:
            topo.append(('pool', 'pool%d' % s))
    return tuple(topo)


def he_init_weights(topology, seed=0, bias_std=0.0):
    """Seeded He-normal weights (std = sqrt(2 / (9 Cin))) in Caffe layout (Cout, Cin, 3, 3).

    The real vgg19.caffemodel is fetched by download_models.sh:3-10 and is not available
    offline; SURVEY.md section 8(d) fixes this synthetic initialisation for every test and bench.
    """
    rng = np.random.RandomState(seed)
    params = OrderedDict()
    for layer in topology:
        if layer[0] != 'conv':
            continue
        _, name, cin, cout = layer
        w = (rng.randn(cout, cin, 3, 3) * np.sqrt(2.0 / (9 * cin))).astype(F32)
        b = (rng.randn(cout) * bias_std).astype(F32) if bias_std else np.zeros(cout, F32)
        params[name] = (w, b)
    return params


def pooled_size(n):
    """Caffe PoolingLayer::Reshape for kernel 2, stride 2, pad 0: ceil((n - 2) / 2) + 1."""
    return int(np.ceil((n - 2) / 2.0)) + 1


# ---------------------------------------------------------------------------------------------
# layer arithmetic (numpy).  x, y are (C, H, W) float32.
# ---------------------------------------------------------------------------------------------

_COL_BUDGET = 48 * 1024 * 1024  # floats per im2col band

# Scratch arrays are kept between calls and the im2col / col2im copies run on a few threads, each on its own channel range.
# Neither changes a single bit of the result (the SGEMM calls and the order of the nine col2im additions per element are the
# same); both only remove time that is not arithmetic: first-touch page faults on hundreds of MB of fresh memory per call and
# single-threaded strided copies, which together were 4/5 of an evaluation at 1024^2.
_SCRATCH = {}
_COPY_THREADS = max(1, min(32, os.cpu_count() or 1))      # (copies are memory-bound: 8 threads here, 32 on the GPU box's 256-core host)
_pool = None


def _scratch(tag, shape):
    n = int(np.prod(shape))
    buf = _SCRATCH.get(tag)
    if buf is None or buf.size < n:
        buf = _SCRATCH[tag] = np.empty(n, F32)
    return buf[:n].reshape(shape)


def _over_channels(fn, channels):
    """fn(c0, c1) over disjoint channel ranges covering [0, channels), on the copy threads (numpy releases the GIL in copies)."""
    global _pool
    parts = min(_COPY_THREADS, channels)
    if parts <= 1:
        fn(0, channels)
        return
    if _pool is None:
        _pool = ThreadPoolExecutor(_COPY_THREADS)
    edges = [channels * i // parts for i in range(parts + 1)]
    list(_pool.map(lambda i: fn(edges[i], edges[i + 1]), range(parts)))


def _band_rows(k_rows, width, height):
    return int(max(1, min(height, _COL_BUDGET // max(1, k_rows * width))))


def conv3x3_forward(x, w, b):
    """Caffe ConvolutionLayer::Forward_cpu: im2col + SGEMM (+ bias); cross-correlation, pad 1."""
    cin, h, wd = x.shape
    cout = w.shape[0]
    xp = _scratch('xp', (cin, h + 2, wd + 2))
    xp[:, 0] = 0; xp[:, -1] = 0; xp[:, :, 0] = 0; xp[:, :, -1] = 0

    def pad(c0, c1):
        xp[c0:c1, 1:-1, 1:-1] = x[c0:c1]
    _over_channels(pad, cin)
    wmat = np.ascontiguousarray(w.reshape(cout, cin * 9))
    out = np.empty((cout, h, wd), F32)
    step = _band_rows(cin * 9, wd, h)
    for r0 in range(0, h, step):
        r1 = min(h, r0 + step)
        col = _scratch('col', (cin, 3, 3, r1 - r0, wd))

        def im2col(c0, c1):
            for ky in range(3):
                for kx in range(3):
                    col[c0:c1, ky, kx] = xp[c0:c1, r0 + ky:r1 + ky, kx:kx + wd]
        _over_channels(im2col, cin)
        if r0 == 0 and r1 == h:
            np.matmul(wmat, col.reshape(cin * 9, h * wd), out=out.reshape(cout, h * wd))
        else:
            band = _scratch('band', (cout, (r1 - r0) * wd))
            np.matmul(wmat, col.reshape(cin * 9, (r1 - r0) * wd), out=band)
            out[:, r0:r1] = band.reshape(cout, r1 - r0, wd)
    out += b.reshape(cout, 1, 1)
    return out


def conv3x3_backward_data(dy, w):
    """Caffe ConvolutionLayer::Backward_cpu, bottom diff only: col = W^T dy, then col2im."""
    cout, h, wd = dy.shape
    cin = w.shape[1]
    wmat_t = np.ascontiguousarray(w.reshape(cout, cin * 9).T)
    dxp = _scratch('dxp', (cin, h + 2, wd + 2))
    dxp[...] = 0
    step = _band_rows(cin * 9, wd, h)
    for r0 in range(0, h, step):
        r1 = min(h, r0 + step)
        top = dy.reshape(cout, h * wd) if (r0 == 0 and r1 == h and dy.flags.c_contiguous) else \
            np.ascontiguousarray(dy[:, r0:r1]).reshape(cout, (r1 - r0) * wd)
        col = _scratch('col', (cin * 9, (r1 - r0) * wd))
        np.matmul(wmat_t, top, out=col)
        col = col.reshape(cin, 3, 3, r1 - r0, wd)

        def col2im(c0, c1):
            for ky in range(3):
                for kx in range(3):
                    dxp[c0:c1, r0 + ky:r1 + ky, kx:kx + wd] += col[c0:c1, ky, kx]
        _over_channels(col2im, cin)
    return np.ascontiguousarray(dxp[:, 1:-1, 1:-1])


def bf16_round(a):
    """fp32 -> bf16 -> fp32 with round-to-nearest-even (no NaN/inf handling: blobs are finite).  Used to
    emulate the bf16 feature path (BASELINE config 3): conv OPERANDS rounded, products/accumulation fp32."""
    u = np.ascontiguousarray(a, F32).view(np.uint32)
    shape = u.shape
    if u.ndim == 0 or u.size < (1 << 16):
        return ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) & np.uint32(0xFFFF0000)).view(F32)
    u = u.reshape(-1, shape[-1])
    r = np.empty_like(u)

    def part(c0, c1):                       # r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000, slab by slab, one temporary
        t = u[c0:c1] >> np.uint32(16)
        t &= np.uint32(1)
        t += np.uint32(0x7FFF)
        t += u[c0:c1]
        np.bitwise_and(t, np.uint32(0xFFFF0000), out=r[c0:c1])
    _over_channels(part, u.shape[0])
    return r.view(F32).reshape(shape)


def _windows(x, fill):
    c, h, w = x.shape
    ho, wo = pooled_size(h), pooled_size(w)
    padded = np.full((c, 2 * ho, 2 * wo), fill, F32)
    padded[:, :h, :w] = x
    return padded.reshape(c, ho, 2, wo, 2).transpose(0, 1, 3, 2, 4).reshape(c, ho, wo, 4)


def maxpool_forward(x):
    """Caffe PoolingLayer::Forward_cpu (MAX): returns (pooled, argmax slot 0..3 in row-major order).

    np.argmax returns the FIRST maximum, which equals Caffe's "strictly greater" scan."""
    win = _windows(x, -np.finfo(F32).max)
    slot = np.argmax(win, axis=-1)
    return np.take_along_axis(win, slot[..., None], axis=-1)[..., 0], slot


def maxpool_backward(dy, slot, in_shape):
    """Caffe PoolingLayer::Backward_cpu (MAX): bottom_diff[argmax] += top_diff."""
    c, h, w = in_shape
    ho, wo = dy.shape[1:]
    win = np.zeros((c, ho, wo, 4), F32)
    np.put_along_axis(win, slot[..., None], dy[..., None], axis=-1)
    full = win.reshape(c, ho, wo, 2, 2).transpose(0, 1, 3, 2, 4).reshape(c, 2 * ho, 2 * wo)
    return np.ascontiguousarray(full[:, :h, :w])


# ---------------------------------------------------------------------------------------------
# the model object (duck type of worker.py:32-106 CaffeModel)
# ---------------------------------------------------------------------------------------------

class NetOracle:
    """Duck type of the reference's ``CaffeModel`` (worker.py:32-106) on the CPU."""

    mean = np.array((123.68, 116.779, 103.939), F32).reshape(3, 1, 1)  # worker.py:34

    def __init__(self, topology=VGG19_TOPOLOGY, params=None, seed=0, full_forward=True, operands='fp32'):
        self.topology = tuple(topology)
        self.params = params if params is not None else he_init_weights(self.topology, seed)
        self.blob_names = ['data'] + [layer[1] for layer in self.topology]
        # Caffe always runs the whole net (worker.py:86).  full_forward=False stops after the
        # deepest requested blob: identical results, used only to time a leaner CPU baseline.
        self.full_forward = full_forward
        # operands='bf16': emulate the bf16 feature path -- a conv whose reduction depth (input channels forward,
        # output channels backward, the 3-channel first layer's data gradient included) is a multiple of 8 sees its
        # activation/diff and weight operands rounded to bf16; everything else (bias, ReLU, pooling, accumulation,
        # the blobs themselves) stays fp32.
        assert operands in ('fp32', 'bf16')
        self.operands = operands
        self._w16 = {}
        self._blobs = {}
        self._slots = {}

    # worker.py:63-66
    def preprocess(self, image):
        chw = np.asarray(image, F32).transpose(2, 0, 1) - self.mean
        return np.ascontiguousarray(chw[None])

    # worker.py:68-71
    def deprocess(self, image):
        return (np.squeeze(image) + self.mean).transpose(1, 2, 0)

    # worker.py:73-75
    def layers(self):
        return list(self.blob_names)

    def blob_shape(self, name, h, w):
        """(C, h, w) of blob `name` for an H x W input."""
        c = 3
        if name == 'data':
            return c, h, w
        for layer in self.topology:
            if layer[0] == 'conv':
                c = layer[3]
            else:
                h, w = pooled_size(h), pooled_size(w)
            if layer[1] == name:
                return c, h, w
        raise KeyError(name)

    # worker.py:77-86
    def forward(self, image, layers=None):
        wanted = self.layers() if layers is None else list(layers)
        last = len(self.topology)
        if not self.full_forward and wanted:
            last = max(self.blob_names.index(n) for n in wanted)
        x = np.ascontiguousarray(image[0], F32)
        self._blobs = {'data': x}
        self._slots = {}
        for layer in self.topology[:last]:
            if layer[0] == 'conv':
                w, b = self.params[layer[1]]
                if self.operands == 'bf16' and layer[2] % 8 == 0:
                    x = conv3x3_forward(bf16_round(x), self._weights16(layer[1]), b)
                else:
                    x = conv3x3_forward(x, w, b)
                np.maximum(x, 0, out=x)            # in-place ReLU: the blob holds post-ReLU data
            else:
                x, self._slots[layer[1]] = maxpool_forward(x)
            self._blobs[layer[1]] = x
            _tick()
        return OrderedDict((n, self._blobs[n][None]) for n in wanted)

    def style_operand(self, name, f2):
        """The features as the style-gradient GEMM S = D @ F sees them: bf16-rounded on the bf16 feature path for the
        blobs whose gradient the engine computes on the bf16 matrix cores (conv outputs with C % 64 == 0), else as is."""
        if self.operands != 'bf16' or f2.shape[0] % 64 != 0:
            return f2
        for layer in self.topology:
            if layer[1] == name:
                return bf16_round(f2) if layer[0] == 'conv' else f2
        return f2

    def gram_operand(self, name, feat):
        """The features (1, C, h, w) as the per-iteration Gram GEMM sees them on the bf16 feature path: bf16-rounded for conv
        blobs with C % 64 == 0 and h w % 64 == 0 (the engine's gram16.hip conditions), else as is."""
        _, c, h, w = feat.shape
        if self.operands != 'bf16' or c % 64 != 0 or (h * w) % 64 != 0:
            return feat
        for layer in self.topology:
            if layer[1] == name:
                return bf16_round(feat) if layer[0] == 'conv' else feat
        return feat

    def _weights16(self, name):
        if name not in self._w16:
            self._w16[name] = bf16_round(self.params[name][0])
        return self._w16[name]

    def adopt_forward_state(self, blobs):
        """Replace the saved forward state (ReLU masks, pool arg-max) by that of another implementation's
        blobs ({name: (1,C,h,w)}) for every blob given.  ReLU and max-pool are discontinuous: two correct fp32
        forwards that differ by rounding (1e-6 relative) disagree on the sign of a handful of pre-activations
        and on a handful of near-tied pooling windows per few million activations, and each such flip changes
        the gradient locally by O(1).  Parity tests of the BACKWARD arithmetic therefore run the oracle's
        backward on the implementation-under-test's own forward state; the end-to-end gradient is compared
        separately with a bound that allows for those flips."""
        for name, arr in blobs.items():
            self._blobs[name] = np.ascontiguousarray(np.asarray(arr, F32)[0])
        for i, layer in enumerate(self.topology):
            if layer[0] == 'pool' and self.blob_names[i] in self._blobs and layer[1] in self._slots:
                _, self._slots[layer[1]] = maxpool_forward(self._blobs[self.blob_names[i]])

    # worker.py:88-106
    def backward(self, diffs):
        """Ranged backward with per-blob diff injection.

        pycaffe semantics being restated (worker.py:92-104; ``start``/``end`` are LAYER names and
        Net::BackwardFromTo is end-inclusive):
          (i)  the diff injected at blob L enters layer L's own backward UNMASKED -- for a conv
               blob the in-place ReLU layer sits after ``start`` and is not executed for it;
               gradient arriving from above IS masked by relu_L (data > 0) before the add;
          (ii) the ``end`` layer runs in two consecutive ranges; the second run overwrites;
          (iii) weight gradients are computed by Caffe and never read -- not computed here.
        Net effect: diff_L = relu_mask_L(incoming) + diffs[L]; then layer L's bottom-diff.
        """
        present = [i for i, n in enumerate(self.blob_names) if n in diffs]
        c, h, w = self._blobs['data'].shape
        if not present:
            return np.zeros((1, c, h, w), F32)
        top = max(present)
        g = None
        for i in range(top, 0, -1):
            layer = self.topology[i - 1]
            name = layer[1]
            if g is not None and layer[0] == 'conv':
                g = g * (self._blobs[name] > 0)
            if name in diffs:
                inj = np.asarray(diffs[name], F32)[0]
                g = inj.copy() if g is None else g + inj
            if layer[0] == 'conv':
                if self.operands == 'bf16' and layer[3] % 8 == 0:
                    g = conv3x3_backward_data(bf16_round(g), self._weights16(name))
                else:
                    g = conv3x3_backward_data(g, self.params[name][0])
            else:
                below = self.blob_names[i - 1]
                g = maxpool_backward(g, self._slots[name], self._blobs[below].shape)
            _tick()
        if 'data' in diffs:
            inj = np.asarray(diffs['data'], F32)[0]
            g = inj.copy() if g is None else g + inj
        return g[None]
