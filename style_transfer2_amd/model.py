"""``HipModel``: the duck type the reference injects into ``StyleTransfer`` (worker.py:121-122),
i.e. ``CaffeModel`` (worker.py:32-106), backed by the gfx950 kernels.

It exists so that the network kernels can be tested in isolation with ANY driver written against
the reference's model interface (the parity tests run the CPU oracle's objective on top of it).
The production loop does not go through these host round trips: ``StyleTransfer`` drives the
engine's device-resident step instead."""

from collections import OrderedDict

import numpy as np

from .engine import Engine

F32 = np.float32


class HipModel:
    mean = np.array((123.68, 116.779, 103.939), F32).reshape(3, 1, 1)   # reference worker.py:34

    def __init__(self, params, topology=None, device=0, engine=None, precision='fp32'):
        self.engine = engine if engine is not None else Engine(topology, device, precision)
        if params is not None:
            self.engine.load_weights(params)

    def preprocess(self, image):
        """reference worker.py:63-66 (host side; the engine has its own device kernel for it)"""
        return np.ascontiguousarray((np.asarray(image, F32).transpose(2, 0, 1) - self.mean)[None])

    def deprocess(self, image):
        """reference worker.py:68-71"""
        return (np.squeeze(image) + self.mean).transpose(1, 2, 0)

    def layers(self):
        """reference worker.py:73-75"""
        return list(self.engine.blob_names)

    def forward(self, image, layers=None):
        """reference worker.py:77-86: returns name -> (1,C,h,w) copies of the requested blobs"""
        wanted = self.layers() if layers is None else list(layers)
        names = self.engine.blob_names
        last = names[max([names.index(n) for n in wanted] + [0])]
        self.engine.forward(image, last)
        return OrderedDict((n, self.engine.get_blob(n)) for n in wanted)

    def backward(self, diffs):
        """reference worker.py:88-106: ranged backward with per-blob diff injection"""
        return self.engine.backward(diffs)
