"""What the web app does to images before they reach the worker, for headless jobs and the benchmark
(SURVEY section 8f item 4): aspect-preserving fit into a square (reference utils.py:210-229), the
uniform-noise initial image (reference app.py:82,251), and a driver that plays the app's message
sequence (app.py:244-262) against a ``StyleTransfer``."""

import numpy as np
from PIL import Image

DEFAULT_WEIGHTS = {'content': {'conv4_2': 0.08},                       # reference initial_weights.yaml:1-3
                   'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1},
                   'deepdream': {}}
DEFAULT_PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}       # reference initial_weights.yaml:4


def fit_into_square(current_size, size, scale_up=False):
    """(w, h) scaled, aspect preserved, so that the longer side equals ``size``."""
    size = int(round(size))
    w, h = current_size
    if not scale_up and max(w, h) <= size:
        return current_size
    if w > h:
        return size, int(round(size * h / w))
    return int(round(size * w / h)), size


def resize_to_fit(image, size, scale_up=True):
    return image.resize(fit_into_square(image.size, size, scale_up), Image.LANCZOS)


def noise_image(hw, seed=None):
    """uint8 uniform noise, like ``np.uint8(np.random.uniform(0, 255, (h, w, 3)))`` in the app."""
    rng = np.random.RandomState(seed) if seed is not None else np.random
    return np.uint8(rng.uniform(0, 255, tuple(hw) + (3,)))


def load_rgb(path):
    return Image.open(path).convert('RGB')


def run_job(transfer, content, style, iterations, size=None, style_size=None, optimizer='adam', step_size=None,
            weights=None, params=None, init=None, seed=0, callback=None):
    """One whole stylisation: content/style are PIL images or HxWx3 arrays.  Returns the final HxWx3 float32
    image.  Mirrors app.init_arrays + 'start': SetImages(reset) -> SetWeights -> SetOptimizer -> iterate."""
    from .device_optimizers import AdamOptimizer, LBFGSOptimizer
    if isinstance(content, Image.Image):
        content = np.uint8(resize_to_fit(content, size or max(content.size)))
    if isinstance(style, Image.Image):
        style = np.uint8(resize_to_fit(style, style_size or size or max(style.size)))
    if init is None:
        init = noise_image(content.shape[:2], seed)
    transfer.set_input(init)
    transfer.set_content(content)
    transfer.set_style(style)
    transfer.reset()
    transfer.set_weights(weights or DEFAULT_WEIGHTS, params or DEFAULT_PARAMS)
    transfer.optimizer_cls = {'adam': AdamOptimizer, 'lbfgs': LBFGSOptimizer}[optimizer]
    transfer.set_step_size(step_size if step_size else {'adam': 10, 'lbfgs': 1}[optimizer])
    transfer.reset()
    if not transfer.start():
        raise RuntimeError('job could not start: inconsistent images')
    image = None
    for i in range(iterations):
        last = i == iterations - 1
        if callback is None and not last:
            transfer.step_async()
        else:
            image, trace = transfer.step()
            if callback is not None:
                callback(i + 1, image, trace)
    return image


def run_tiled_job(net_params, content, style, iterations, grid, size=None, style_size=None, weights=None, params=None, init=None,
                  seed=0, device=0, precision='fp32', topology=None, callback=None, optimizer='adam', step_size=None):
    """The same job for an image no single engine holds (an 8192 x 8192 image has conv1 blobs of 17 GB each): the image is cut into
    grid = (rows, cols) tiles, every tile + apron is an engine context of THIS process on the one GPU (tiled.InProcessFabric: one
    thread per rank, the all-reduces and strip exchanges are device-to-device copies); one st_tile_step per rank and iteration, Adam or
    L-BFGS (the Gram form: one all-reduce of the new inner products per step).
    Same result as run_job where both can run (tests/test_gpu_jobs.py).  Image edges must be multiples of 16 * rows / cols
    (tiling.TileGrid).  Returns the stitched HxWx3 float32 image; callback(i, trace values) after every iteration."""
    from . import tiled, tiling
    from .engine import VGG19_TOPOLOGY
    from .tile_backend import HipTileBackend
    topology = topology or VGG19_TOPOLOGY
    if isinstance(content, Image.Image):
        content = np.uint8(resize_to_fit(content, size or max(content.size)))
    if isinstance(style, Image.Image):
        style = np.uint8(resize_to_fit(style, style_size or size or max(style.size)))
    if init is None:
        init = noise_image(content.shape[:2], seed)
    weights, params = weights or DEFAULT_WEIGHTS, params or DEFAULT_PARAMS
    rows, cols = grid
    h, w = content.shape[:2]
    deepest = max(i for i, layer in enumerate(topology) if any(layer[1] in weights[k] and weights[k][layer[1]] for k in weights))
    tg = tiling.TileGrid(h, w, rows, cols, topology, deepest + 1)
    world = rows * cols
    fabric = tiled.InProcessFabric(world, timeout=600.0)
    ranks, backends = [], []
    for r in range(world):
        b = HipTileBackend(net_params, tg, r, content, style, init, weights, params, step_size=step_size or {'adam': 10, 'lbfgs': 1}[optimizer],
                           topology=topology, device=device, precision=precision, optimizer=optimizer)
        backends.append(b)
        b.comm_init_local(r, world, fabric)
        ranks.append(tiled.FusedTiledTransfer(tg, r, b))
    try:
        for i in range(iterations):
            vals = tiled.run_in_process(ranks, 1, fabric)[0][0]
            if callback is not None:
                callback(i + 1, vals)
        image = np.zeros((h, w, 3), np.float32)
        for r in range(world):
            t = tg.tiles[r]
            image[t.y0:t.y1, t.x0:t.x1] = ranks[r].tile_image()
    except RuntimeError as err:
        if getattr(err, 'still_running', False):     # a rank thread is still inside the engine: freeing its context under it would
            backends = []                            # pull device memory from running kernels -- leak the contexts and report
        raise
    finally:
        for b in backends:
            b.engine.close()
    return image
