#!/usr/bin/env python3
"""Benchmark of the style-transfer inner loop on MI355X (BASELINE.json: iterations/sec @1024px VGG19; images/hour at
1/2/4/8 GPUs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 1024] [--optimizer adam] [--precision fp32]
    python bench.py --size 2048 --optimizer lbfgs --precision bf16          # BASELINE configs[2]
    python bench.py --examples                                              # BASELINE configs[0]: the example pair at 256 px, 50 Adam iterations
    python bench.py --gpus 8 --tiled 2x4 --size 8192                        # BASELINE configs[4] (tile-sharded image)
    python bench.py --tiled 2x4 --size 8192 --steps 5                       # the same job with its eight ranks on ONE GPU (time-sliced)

One "step" = one ``StyleTransfer.step()`` = forward to conv5_1, 1 content + 5 style loss terms, ranged backward,
TV / p-norm and the optimizer update (reference worker.py:303-310).  Inputs are synthetic (seeded He-normal VGG19
weights, uniform-noise uint8 images: SURVEY section 8d) and resident in HBM before the timed region; nothing is read
back inside it.

N > 1: jobs are the unit the reference scales by (one worker per `gpu` key: config.ini:10, worker.py:328,
router.py:67-84), so every rank runs an independent job on its own GPU with NO data-path collective; `value` is the
aggregate it/s over all ranks and scaling is weak.  The ranks come either from the launcher (torch.distributed.run
sets RANK / LOCAL_RANK / WORLD_SIZE) or, when those are absent and --gpus N > 1, from this script itself: it starts N
fresh child processes (one per GPU) BEFORE anything touches the GPU and relays rank 0's line.

Rank 0 prints ONE JSON line.  `value` comes from the median of R back-to-back timed blocks of exactly K steps (each
bracketed by barrier + device sync, max over ranks); all blocks are listed under `timing`.  `roofline` is for the
dominant kernel class (the conv3x3 launches on the matrix cores): `achieved` counts the FLOPs the MFMA pipe executes
(Winograd launches perform 4/9 of the direct-convolution FLOPs), so `frac` is a true fraction of the MFMA peak; the
algorithmic (direct-convolution) figure is kept beside it.  `cpu_baseline` (kind "port") and `parity` come
from the CPU oracle run on this host on the same inputs, N = 1 only: the objective at the initial image (loss, gradient, ReLU
sign census) and the ITERATE after the oracle's timed step(s) (`parity.image_mse`, 0..255 units).  `worker_level` is the rate
including one `Iterate` per step, which `value` excludes: D2H + pickle on one thread / with the sender thread / pipelined, and the
worker's zero-copy form (the GPU copies the iterate into a pre-formatted pickle in pinned memory; the transport gets that buffer).
"""

import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E
PEAK_BF16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (8 passes x 4 cyc)
SPLIT_EXECUTED = 6.0 * 4.0 / 9.0   # split-operand Winograd: bf16 FLOPs the pipe executes per direct-convolution FLOP (six partial products, 4/9 of the multiplies)
SPLIT_CLASSES = ('conv3x3_fwd_wino_split_bf16x6', 'conv3x3_dgrad_wino_split_bf16x6')
SPLIT_DTYPE = ('f32 results; conv products as 6 exact bf16 partial products of three-way split f32 operands on the bf16 matrix cores, f32 accumulate; '
               'everything else f32')
WEIGHTS = {'content': {'conv4_2': 0.08},
           'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1},
           'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
ITERS_PER_IMAGE = 500
STEP_SIZES = {'adam': 10, 'lbfgs': 1}


def images(size):
    rs = np.random.RandomState
    shape = (size, size, 3)
    return (rs(1).randint(0, 256, shape).astype(np.uint8), rs(2).randint(0, 256, shape).astype(np.uint8),
            rs(3).randint(0, 256, shape).astype(np.uint8))


EXAMPLES_FIT = 256
EXAMPLES_ITERS = 50


def example_inputs():
    """BASELINE configs[0]: the reference's example pair as app.py would have sent it (app.py:82,244-262): decoded pixels
    (tests/golden/config1_sources.npz, written from the reference's examples/*.jpg by tests/golden/make_golden.py) ->
    jobs.resize_to_fit(256) (reference utils.py:210-229; checked here against the reference's own output,
    tests/golden/config1_inputs.npz) -> uniform-noise initial image."""
    from PIL import Image
    from style_transfer2_amd import jobs
    gold = os.path.join(HERE, 'tests', 'golden')
    src, ref = np.load(os.path.join(gold, 'config1_sources.npz')), np.load(os.path.join(gold, 'config1_inputs.npz'))
    content = np.uint8(jobs.resize_to_fit(Image.fromarray(src['golden_gate']), EXAMPLES_FIT))
    style = np.uint8(jobs.resize_to_fit(Image.fromarray(src['starry_night']), EXAMPLES_FIT))
    same = bool(np.array_equal(content, ref['golden_gate']) and np.array_equal(style, ref['starry_night']))
    return content, style, jobs.noise_image(content.shape[:2], seed=0), same


def job_inputs(args):
    """(content, style, init, weights, params) of the benched workload."""
    if args.examples:
        from style_transfer2_amd import jobs
        content, style, init, _ = example_inputs()
        return content, style, init, jobs.DEFAULT_WEIGHTS, jobs.DEFAULT_PARAMS
    return images(args.size) + (WEIGHTS, PARAMS)


def make_job(inputs, optimizer, device, precision='fp32', conv_algo=1):
    import style_transfer2_amd as st2
    from style_transfer2_amd import weights as st2_weights
    content, style, init, weights, params = inputs
    model = st2.HipModel(st2_weights.he_normal(st2.VGG19_TOPOLOGY, seed=0), device=device, precision=precision)
    if conv_algo != 1:
        model.engine.set_conv_algo(conv_algo)      # 2: Winograd-domain products as six bf16 partial products of split operands (fp32 results)
    job = st2.StyleTransfer(model)
    job.set_weights(weights, params)         # first: the engine then keeps content features only where a content weight reads them
    job.set_input(init)
    job.set_content(content)
    job.set_style(style)
    job.optimizer_cls = {'adam': st2.AdamOptimizer, 'lbfgs': st2.LBFGSOptimizer}[optimizer]
    job.set_step_size(STEP_SIZES[optimizer])
    job.reset()
    assert job.start()
    return job


class StubJob:
    """Stands in for a job in the CPU test of the multi-rank plumbing (tests/test_bench_fanout_cpu.py): a step is a
    1-ms sleep, there is no engine.  Never used on a GPU box."""
    engine = None

    def step_async(self):
        time.sleep(0.001)

    def sync(self):
        pass


def workload_label(args):
    if args.examples:
        return ('configs[0]: examples golden_gate + starry_night fitted to %d px (content 192x256, style 160x256) by jobs.resize_to_fit, '
                'uniform-noise init, VGG19 (seeded weights), initial_weights.yaml losses, adam fp32, %d iterations per image'
                % (EXAMPLES_FIT, args.examples_iters))
    what = '%dx%d single image per GPU, VGG19 to conv5_1, content conv4_2 + 5 style layers, %s %s, %d iterations per image' % (
        args.size, args.size, args.optimizer, args.precision, ITERS_PER_IMAGE)
    if getattr(args, 'conv_algo', 1) != 1:
        tag = 'custom (conv algorithm %d: %s)' % (args.conv_algo, 'direct kernel only' if args.conv_algo == 0 else 'split-operand Winograd on the bf16 matrix cores, fp32 results')
    elif (args.size, args.optimizer, args.precision) == (1024, 'adam', 'fp32'):
        tag = 'configs[1]' if args.gpus == 1 else 'configs[3] (%d independent configs[1] jobs)' % args.gpus
    elif (args.size, args.optimizer, args.precision) == (2048, 'lbfgs', 'bf16'):
        tag = 'configs[2]'
    else:
        tag = 'custom (not a BASELINE.json config)'
    return '%s: %s' % (tag, what)


def class_roofline(name, rec, conv_peak):
    """One kernel class against its rooflines: TFLOP/s vs the MFMA peak of its operand type when the engine recorded
    algorithmic flops for it (Winograd classes: the 4/9 of them that the pipe executes), GB/s vs HBM when it recorded
    algorithmic bytes (both for classes that have both)."""
    sec = rec['ms'] * 1e-3
    out = {}
    if rec['flops'] > 0:
        split = 'wino_split' in name             # six bf16 partial products per Winograd-domain product, on the bf16 pipe
        executed = rec['flops'] * (SPLIT_EXECUTED if split else 4.0 / 9.0 if 'wino' in name else 1.0)
        tf = executed / sec / 1e12
        out['TFLOP/s'] = round(tf, 1)
        out['frac_mfma'] = round(tf / (PEAK_BF16_MFMA_TFLOPS if (split or name.endswith('bf16')) else PEAK_F32_MFMA_TFLOPS), 3)
        if 'wino' in name:
            out['algorithmic_TFLOP/s'] = round(rec['flops'] / sec / 1e12, 1)
    if rec['bytes'] > 0:
        gbs = rec['bytes'] / sec / 1e9
        out['GB/s'] = round(gbs, 1)
        out['frac_hbm'] = round(gbs / PEAK_HBM_GBS, 3)
    return out


def source_hash():
    """sha256[:16] over the kernel / engine sources: the committed PMC summaries carry the hash they were measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(HERE, 'style_transfer2_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.cpp', '.h')):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), 'rb').read())
    return h.hexdigest()[:16]


def committed_pmc(name, args):
    """A committed rocprofv3 --pmc summary of this same command (tools/profile_round.sh), or (None, why).  The counters
    need rocprofv3 around the process, so they cannot be taken inside this run: the line names the file and says
    whether it was measured on the kernel sources that are running now."""
    path = os.path.join(HERE, 'profiles', name)
    if (args.size, args.optimizer, args.precision) != (1024, 'adam', 'fp32'):
        return None, 'no committed PMC pass for this configuration'
    if not os.path.exists(path):
        return None, 'profiles/%s absent' % name
    try:
        data = json.load(open(path))
    except ValueError:
        return None, 'profiles/%s unreadable' % name
    meta = data.get('_meta', {})
    if meta.get('source_sha16') != source_hash():
        return None, 'profiles/%s was measured on other kernel sources (%s, now %s): stale, not reported' % (
            name, meta.get('source_sha16'), source_hash())
    return data, 'profiles/%s (rocprofv3 --pmc pass of this command, %s, same kernel sources)' % (name, meta.get('profile', '?'))


# ------------------------------------------------------------------------------------------------ CPU oracle legs
def cpu_baseline_and_parity(inputs, optimizer, precision, dev_eval, iterations=1):
    """The CPU oracle ("port") on this host: setup, one objective evaluation at the initial image (untimed; it captures
    the norms and is what `parity` compares with the HIP path's evaluation of the same state), then `iterations` timed steps
    whose last iterate is compared with the HIP path's iterate after the same steps from the same state."""
    import oracle
    from threadpoolctl import threadpool_info
    oracle.keep_freed_memory()
    content, style, init, weights, params = inputs
    topo = oracle.VGG19_TOPOLOGY
    net = oracle.NetOracle(topo, oracle.he_init_weights(topo, seed=0), full_forward=False,
                           operands='bf16' if precision == 'bf16' else 'fp32')
    job = oracle.TransferOracle(net)
    # (content features / style Grams of the weighted blobs only -- the reference keeps all 22 because the weights may change later;
    #  here they are fixed: same values, two whole-net forwards less of untimed set-up)
    job.feature_layers = [n for n in net.layers() if any(abs(weights[k].get(n, 0)) > 1e-15 for k in weights)]
    job.set_input(init)
    job.set_content(content)
    job.set_style(style)
    job.reset()
    job.set_weights(weights, params)
    job.set_optimizer(optimizer, STEP_SIZES[optimizer])
    job.start()
    # (what the objective hands to its ranged backward, and what came back: the same diffs are back-propagated once more below on the
    # HIP path's branch decisions)
    rec, plain_backward = {}, net.backward

    def recording_backward(diffs):
        rec['diffs'] = dict(diffs)
        rec['scd'] = plain_backward(diffs)
        return rec['scd']
    net.backward = recording_backward
    loss0, grad0 = job.opfunc(job.input)
    net.backward = plain_backward
    parity = None
    if dev_eval is not None:
        from oracle.caffe_net import maxpool_forward
        from oracle.receptive import receptive_geometry, paint_receptive_fields
        ld, gd, signs = dev_eval[:3]
        slots = dev_eval[4] if len(dev_eval) > 4 else {}
        g64, d64 = grad0.astype(np.float64), gd.astype(np.float64)
        flips = total = pool_flips = 0
        # where the two forwards took different branches (ReLU sign per conv blob, first-maximum slot per pooling window), and the
        # image-space receptive fields of those units: the gradient is held to BASELINE.md section 3's gate OUTSIDE them
        names = ['data'] + [layer[1] for layer in topo]
        geo = receptive_geometry(topo)
        mask = np.zeros(grad0.shape[2:], bool)
        for name, packed in signs.items():
            blob = net._blobs[name]
            diff = np.unpackbits(np.packbits(blob > 0) ^ packed)[:blob.size].reshape(blob.shape).astype(bool)
            flips += int(diff.sum())
            total += blob.size
            if diff.any():
                paint_receptive_fields(mask, np.argwhere(diff.any(axis=0)), geo[names.index(name)])
        for name, slot in slots.items():
            if name in net._slots:
                diff = net._slots[name] != slot
                pool_flips += int(diff.sum())
                if diff.any():
                    paint_receptive_fields(mask, np.argwhere(diff.any(axis=0)), geo[names.index(name)])
        pix = np.abs(gd - grad0)[0].max(0)
        out = ~mask
        outside = (float(np.linalg.norm((d64 - g64)[0][:, out]) / np.linalg.norm(g64[0][:, out])) if out.any() else None)
        # BASELINE.md section 3's gate over EVERY pixel: the oracle's own backward (same injected diffs) re-run with the HIP path's ReLU
        # signs and first-maximum slots adopted -- nothing else changes -- must agree with the HIP gradient to 1e-4
        adopted = None
        if 'diffs' in rec and len(signs) == sum(1 for layer in topo[:17] if layer[0] == 'conv'):
            kept = {n: net._blobs[n] for n in signs}
            kept_slots = dict(net._slots)
            for n, packed in signs.items():
                shape = net._blobs[n].shape
                net._blobs[n] = np.unpackbits(packed)[:int(np.prod(shape))].reshape(shape).astype(np.float32)      # 1 / 0: only `> 0` is read
            for n, slot in slots.items():
                if n in net._slots:
                    net._slots[n] = slot
            g_adopt = (grad0 - rec['scd'] + plain_backward(rec['diffs'])).astype(np.float64)
            adopted = float(np.linalg.norm(d64 - g_adopt) / np.linalg.norm(g_adopt))
            net._blobs.update(kept)
            net._slots.update(kept_slots)
        whole = float(np.linalg.norm(d64 - g64) / np.linalg.norm(g64))
        parity = {'against': 'CPU oracle (%s conv operands), objective at the initial image, same inputs' % precision,
                  'loss_rel': float(abs(float(ld) - float(loss0)) / abs(float(loss0))),
                  'grad_rel_l2': whole,
                  'grad_rel_l2_branch_decisions_adopted': adopted,
                  'grad_rel_l2_outside_flipped_fields': outside,
                  'receptive_field_union_frac': float(mask.mean()),
                  'grad_cosine': float(np.vdot(d64, g64) / (np.linalg.norm(d64) * np.linalg.norm(g64))),
                  'relu_sign_flips': flips, 'pool_argmax_flips': pool_flips, 'activations': total,
                  'affected_pixel_frac': float(np.mean(pix > 1e-3 * np.abs(grad0).max())),
                  'gate': 'BASELINE.md section 3: gradient rel-L2 <= 1e-4 per step, every pixel',
                  'gate_met_by_grad_rel_l2': bool(whole <= 1e-4),
                  'gate_met_with_branch_decisions_adopted': (bool(adopted <= 1e-4) if adopted is not None else None),
                  'note': 'ReLU / max-pool are discontinuous: two correct fp32 forwards take different branches at a few dozen of 3e8 '
                          'activations, and each flip changes the gradient by O(1) inside the image-space receptive field of the flipped '
                          'unit.  grad_rel_l2 (raw, every pixel) is over the gate for that reason alone: with the HIP path\'s %d branch '
                          'decisions adopted by the oracle (its own backward, its own injected diffs, nothing else changed) the gradient '
                          'agrees at EVERY pixel (grad_rel_l2_branch_decisions_adopted); outside the receptive fields of the flipped units '
                          'it agrees without that (grad_rel_l2_outside_flipped_fields).  BASELINE.md section 3 carries the same qualification.'
                          % (flips + pool_flips)}
    extra = 1 if (optimizer == 'lbfgs' and iterations == 1) else 0
    if extra:
        job.step()                  # the first L-BFGS step costs two evaluations; time a steady-state one
    samples = []
    for _ in range(iterations):
        t0 = time.perf_counter()
        image, trace = job.step()
        samples.append(time.perf_counter() - t0)
    dt = sum(samples)
    if parity is not None and len(dev_eval) > 3 and dev_eval[3] is not None:
        dev_image, dev_trace = dev_eval[3]
        # the iterate itself (reference worker.py:303-310 returns deprocess(x)): 0..255 units
        parity['image_mse'] = float(np.mean((np.asarray(dev_image, np.float64) - np.asarray(image, np.float64)) ** 2))
        parity['image_max_abs'] = float(np.max(np.abs(np.asarray(dev_image, np.float64) - image)))
        parity['image_after'] = '%d %s iteration(s) from the same initial state on both sides' % (iterations + extra, optimizer)
        parity['step_loss_rel'] = float(abs(dev_trace['loss'] - trace['loss']) / abs(trace['loss']))
    pools = threadpool_info()
    threads = max([p.get('num_threads', 1) for p in pools] + [1])
    base = {'value': iterations / dt, 'unit': 'it/s', 'cores': threads, 'host_cpus': os.cpu_count(), 'kind': 'port',
            'blas': [{k: p.get(k) for k in ('internal_api', 'version', 'threading_layer', 'architecture', 'num_threads')} for p in pools],
            'thread_placement': 'no pinning: %d CPUs in this process\'s affinity mask, OPENBLAS/OMP thread-count variables %s'
                                % (len(os.sched_getaffinity(0)), {k: os.environ[k] for k in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS') if k in os.environ} or 'unset'),
            'iterations': iterations, 'it_s_min_max': [1.0 / max(samples), 1.0 / min(samples)],
            'sample': '%d %s iteration(s) at %dx%d, timed one by one (after 1 untimed objective evaluation), numpy+OpenBLAS oracle, forward stops at the deepest weighted layer'
                      % (iterations, optimizer, init.shape[0], init.shape[1])}
    return base, parity


def device_eval_for_parity(job, iterations=1):
    """Objective at the initial image on the HIP path, plus the ReLU sign pattern (bit-packed) of every conv blob; then the
    iterate after `iterations` steps from that state (what the oracle leg does after its own evaluation)."""
    loss, grad = job.opfunc()
    eng = job.engine
    signs = {}
    from style_transfer2_amd import StError
    for layer in eng.topology[:17]:
        if layer[0] == 'conv':
            try:
                signs[layer[1]] = np.packbits(eng.get_blob(layer[1])[0] > 0)
            except StError:             # bf16 lean data flow: this blob exists only as a bf16 copy -- not part of the census
                pass
    # first-maximum slot of every pooling window (one byte each), from the conv blob the pool reads
    slots = {}
    names = [layer[1] for layer in eng.topology[:17]]
    for i, layer in enumerate(eng.topology[:17]):
        if layer[0] == 'pool' and i >= 1 and names[i - 1] in signs:
            try:
                from oracle.caffe_net import maxpool_forward       # (the parity leg is the checker: bench.py's cpu_baseline leg only)
                slots[layer[1]] = maxpool_forward(eng.get_blob(names[i - 1])[0])[1]
            except StError:
                pass
    stepped = None
    for _ in range(iterations):
        stepped = job.step()
    return loss, grad, signs, stepped, slots


def worker_level(job, steps):
    """Rate with one Iterate per step as worker.py sends it (SURVEY 8d "report both"): image D2H over PCIe + the sending side of
    pyzmq's send_pyobj (pickle of the HxWx3 float32 image) on one thread / with worker.AsyncSender / pipelined; and the worker's
    zero-copy form, where the GPU copies the iterate into a pre-formatted pickle in pinned memory and the transport is handed
    that buffer (no host-side copy).  The sinks stand where libzmq would: `handoff` takes the buffer and returns, `tcp` pushes
    every byte through a loopback TCP socket drained by a reader thread (the app's side of the wire runs on this host too)."""
    import pickle
    import socket as socket_mod
    import threading
    import messages
    import worker as worker_mod

    class PickleSink:
        def __init__(self):
            self.bytes = 0

        def send_pyobj(self, obj):
            self.bytes += len(pickle.dumps(obj, protocol=pickle.DEFAULT_PROTOCOL))

        def send(self, data, copy=True, track=False):          # zero-copy hand-off: libzmq would queue the buffer itself
            self.bytes += len(data)

    class TcpSink(PickleSink):
        """Every frame crosses a loopback TCP connection; a reader thread drains it (kernel copies on both sides)."""
        def __init__(self):
            super().__init__()
            srv = socket_mod.socket()
            srv.bind(('127.0.0.1', 0))
            srv.listen(1)
            self.out = socket_mod.create_connection(srv.getsockname())
            self.inp, _ = srv.accept()
            srv.close()
            self.reader = threading.Thread(target=self._drain, daemon=True)
            self.reader.start()

        def _drain(self):
            buf = bytearray(8 << 20)
            while self.inp.recv_into(buf):
                pass

        def send(self, data, copy=True, track=False):
            self.out.sendall(data)
            self.bytes += len(data)

        def close(self):
            self.out.close()
            self.reader.join(10)
            self.inp.close()

    out = {}
    n = max(steps, 30)
    sink = PickleSink()

    def leg(mode, sink=sink):
        framed = mode.startswith('zero_copy')
        if mode == 'pipelined' or framed:
            for _ in range(2):                       # (the first begin allocates the rotating pinned buffers)
                job.step_begin()
            while job.steps_pending:
                job.step_end(copy=False)
        else:
            for _ in range(2):
                job.step_async()
        job.engine.sync()
        send = sink if mode == 'sync' else worker_mod.AsyncSender(sink)

        def collect():
            if framed:
                send.send_frame(job.step_end(frame=True)[3])
            else:
                image, trace, index = job.step_end(copy=False)       # the sender thread pickles the pinned view
                send.send_pyobj(messages.Iterate(image, index, trace))
        t0 = time.perf_counter()
        if mode == 'pipelined' or framed:
            # the worker's default loop: iteration k + 1 is begun before iterate k is collected (st_step_begin / st_step_end)
            for _ in range(n):
                job.step_begin()
                if job.steps_pending > 1:
                    collect()
            while job.steps_pending:
                collect()
        else:
            for _ in range(n):
                image, trace = job.step()
                send.send_pyobj(messages.Iterate(image, job.t, trace))
            job.engine.sync()
        if mode != 'sync':
            send.close()
        return n / (time.perf_counter() - t0)

    modes = ('sync', 'async') + (('pipelined',) if hasattr(job, 'step_begin') else ())
    for mode in modes:
        rates = sorted(leg(mode) for _ in range(3))
        out[mode + '_iterate_it_s'] = rates[1]          # median of three legs of n iterations
        out[mode + '_iterate_it_s_min_max'] = [rates[0], rates[2]]
    out['iterate_MB'] = sink.bytes / (3 * len(modes) * n) / 1e6
    if hasattr(job, 'enable_iterate_frames'):
        job.enable_iterate_frames()                      # from here on the pinned slots carry room for the pickle around the image
        rates = sorted(leg('zero_copy') for _ in range(3))
        out['zero_copy_iterate_it_s'] = rates[1]
        out['zero_copy_iterate_it_s_min_max'] = [rates[0], rates[2]]
        tcp = TcpSink()
        out['zero_copy_tcp_loopback_it_s'] = leg('zero_copy_tcp', tcp)
        tcp.close()
    out['steps'] = n
    out['note'] = ('one Iterate (D2H + pickle) per step: on one host thread / with the sender thread / with the sender thread and the next '
                   'iteration begun before the iterate is collected; zero_copy = the worker default on its own pyzmq sockets: the GPU '
                   'copies the iterate into a pre-formatted pickle in pinned memory and the transport is handed that buffer '
                   '(tcp_loopback: every byte then crosses a loopback TCP socket to a reader thread on this host); never `value`')
    return out


def conv_class(prof, f32):
    """The dominant kernel class of a profiled leg: every conv3x3 launch on the matrix cores.  Returns (direct records, Winograd
    records, executed TFLOP/s, algorithmic TFLOP/s, ms, launches): the engine records ALGORITHMIC flops per launch (direct
    convolution); a Winograd launch executes 4/9 of them on the MFMA pipe."""
    direct = [prof[k] for k in ('conv3x3_fwd_mfma_f32', 'conv3x3_dgrad_mfma_f32', 'conv3x3_fwd_mfma_bf16', 'conv3x3_dgrad_mfma_bf16') if k in prof]
    wino = [prof[k] for k in ('conv3x3_fwd_wino_f32', 'conv3x3_dgrad_wino_f32') if k in prof]
    # (launches of the split-operand Winograd kernel run on the OTHER matrix pipe: split_conv_class prices them; not part of this class)
    ms = sum(c['ms'] for c in direct + wino)
    sec = ms * 1e-3
    flops = sum(c['flops'] for c in direct + wino)
    executed = (sum(c['flops'] for c in direct) + sum(c['flops'] for c in wino) * 4.0 / 9.0) / sec / 1e12 if ms else 0.0
    return direct, wino, executed, (flops / sec / 1e12 if ms else 0.0), ms, sum(c['launches'] for c in direct + wino)


def split_conv_class(prof):
    """The launches of the split-operand Winograd kernel (st_set_conv_algo 2) against the bf16 MFMA peak: executed = 6 x 4/9 of the
    direct-convolution FLOPs the engine records.  None when none ran."""
    recs = [prof[k] for k in SPLIT_CLASSES if k in prof]
    ms = sum(c['ms'] for c in recs)
    if not recs or ms <= 0:
        return None
    flops = sum(c['flops'] for c in recs)
    launches = sum(c['launches'] for c in recs)
    executed = flops * SPLIT_EXECUTED / (ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'kernel': 'conv3x3 Winograd F(2x2,3x3) with split operands on the bf16 matrix cores (%d launches)' % launches,
            'achieved': executed, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': executed / PEAK_BF16_MFMA_TFLOPS,
            'achieved_is': 'bf16 FLOPs the MFMA pipe executes (6 partial products x 4/9 of the direct-convolution FLOPs) / kernel time (HIP events on the engine stream)',
            'algorithmic': flops / (ms * 1e-3) / 1e12, 'ms': ms, 'launches': launches}


def profiled_leg(job, steps):
    job.engine.profile_enable(True)
    for _ in range(steps):
        job.step_async()
    prof = job.engine.profile_read()
    job.engine.profile_enable(False)
    return prof


def extra_config_leg(label, inputs, optimizer, precision, steps, blocks, warmup, device, conv_algo=1):
    """One more BASELINE config measured in the default run, reported under `extra_configs` (never `value`): the same
    timed-region contract (warm-up, then `blocks` blocks of exactly `steps` device-resident steps, median), plus the conv class's
    fraction of its MFMA peak from a short profiled leg."""
    from style_transfer2_amd import distributed as st2_dist
    t0 = time.perf_counter()
    job = make_job(inputs, optimizer, device, precision, conv_algo)
    solo = st2_dist.Group.__new__(st2_dist.Group)
    solo.rank, solo.local_rank, solo.world, solo.dist, solo.device = 0, 0, 1, None, None
    try:
        times = [st2_dist.timed_region(solo, job.step_async, steps, warmup if b == 0 else 0, job.engine.sync) for b in range(blocks)]
        elapsed = statistics.median(times)
        f32 = precision == 'fp32'
        psteps = max(3, min(10, steps))
        prof = profiled_leg(job, psteps)
        _, wino, executed, algorithmic, ms, launches = conv_class(prof, f32)
        peak = PEAK_F32_MFMA_TFLOPS if f32 else PEAK_BF16_MFMA_TFLOPS
        h, w = inputs[2].shape[:2]
        out = {'workload': label, 'value': steps / elapsed, 'unit': 'it/s', 'ms_per_step': 1e3 * elapsed / steps, 'steps': steps, 'blocks': blocks,
               'warmup': warmup, 'block_ms': [round(1e3 * t, 3) for t in times], 'size': [h, w], 'optimizer': optimizer,
               'dtype': SPLIT_DTYPE if conv_algo == 2 else 'f32' if f32 else 'bf16 conv operands, f32 accumulate/Gram/optimizer',
               'roofline': {'bound': 'mfma', 'kernel': 'conv3x3 on the %s matrix cores' % ('f32' if f32 else 'bf16'), 'achieved': executed,
                            'peak': peak, 'unit': 'TFLOP/s', 'frac': executed / peak, 'algorithmic': algorithmic,
                            'flops_are': 'SURVEY 8(d): 2*9*K*M*H*W per conv launch, nothing else (style-gradient chunks fused into a bf16 data '
                                         'gradient are booked to style_grad_fused_in_conv_dgrad_bf16)',
                            'conv_ms_per_step': ms / psteps, 'launches_per_step': launches / psteps},
               'kernel_ms_per_step': {k: round(v['ms'] / psteps, 4) for k, v in sorted(prof.items())}}
        fused = prof.get('style_grad_fused_in_conv_dgrad_bf16')
        if fused:
            out['roofline']['fused_style_gradient_gflop_per_step'] = fused['flops'] / psteps / 1e9
            out['roofline']['frac_with_fused_style_flops'] = (executed + fused['flops'] / (ms * 1e-3) / 1e12) / peak if ms else None
        split = split_conv_class(prof)
        if split:
            # the launches the split kernel took are priced against the bf16 peak; what stayed on the fp32 pipe (conv1_1, shapes the
            # kernel cannot take) keeps the line above
            split['conv_ms_per_step'] = split.pop('ms') / psteps
            split['launches_per_step'] = split.pop('launches') / psteps
            out['roofline_f32_pipe_rest'] = out['roofline']
            out['roofline'] = split
        out['leg_seconds'] = time.perf_counter() - t0
        return out
    finally:
        job.engine.close()


def extra_configs(args, device):
    """BASELINE configs[2], [0] and (memory permitting) [4]-on-one-GPU as short legs of the default N = 1 run, so that the driver's one
    command sees every config; each failure is recorded as {'error': ...} and never disturbs the headline line."""
    out = {}
    t_all = time.perf_counter()

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as err:          # noqa: BLE001  (an extra leg must never cost the headline line)
            out[name] = {'error': '%s: %s' % (type(err).__name__, err)}
    guarded('configs[2] 2048x2048 lbfgs bf16', lambda: extra_config_leg(
        'configs[2]: 2048x2048 single image, L-BFGS step 1, bf16 conv operands / fp32 accumulate + Gram + optimizer',
        images(2048) + (WEIGHTS, PARAMS), 'lbfgs', 'bf16', 10, 3, 5, device))

    guarded('configs[1] split-operand convs', lambda: extra_config_leg(
        'configs[1]\'s job (1024x1024, VGG19 to conv5_1, content conv4_2 + 5 style layers, Adam step 10) with st_set_conv_algo(ctx, 2): every '
        'eligible conv as Winograd F(2x2,3x3) whose transform-domain products are six exact bf16 partial products of three-way split fp32 '
        'operands (bf16 matrix cores, fp32 accumulate); fp32 results under the fp32 parity bars (tests/test_gpu_wino_split.py, '
        'tests/test_gpu_fullsize.py) -- NOT the headline arithmetic: `value` above is the IEEE-fp32 path',
        images(1024) + (WEIGHTS, PARAMS), 'adam', 'fp32', 20, 3, 5, device, conv_algo=2))

    def examples():
        from style_transfer2_amd import jobs
        content, style, init, same = example_inputs()
        leg = extra_config_leg('configs[0] on the device: examples golden_gate + starry_night fitted to %d px, noise init, Adam step 10, '
                               '%d iterations per image (the CPU-oracle leg of this config: bench.py --examples)' % (EXAMPLES_FIT, EXAMPLES_ITERS),
                               (content, style, init, jobs.DEFAULT_WEIGHTS, jobs.DEFAULT_PARAMS), 'adam', 'fp32', EXAMPLES_ITERS, 3, 5, device)
        leg['images_per_hour'] = leg['value'] * 3600.0 / EXAMPLES_ITERS
        leg['resize_to_fit_matches_reference_fixture'] = same
        return leg
    guarded('configs[0] examples 256px adam fp32', examples)

    def tiled_8192():
        # eight engine contexts of 24.7 GB each, time-sliced on this one GPU; a child process (torch's HIP runtime must come up first
        # there: tools/bench_tiled_one_gpu.py), which declines by itself when less than 230 GB of HBM is free
        cmd = [sys.executable, os.path.join(HERE, 'tools', 'bench_tiled_one_gpu.py'), '--size', '8192', '--grid', '2x4', '--steps', '3',
               '--warmup', '1', '--need-free-gib', '230']
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
        except subprocess.TimeoutExpired as err:        # (the child is killed with GPU work in flight: say so, with what it printed)
            return {'error': 'tools/bench_tiled_one_gpu.py did not finish within 240 s and was killed',
                    'stderr_tail': (err.stderr or b'').decode(errors='replace')[-1500:]}
        lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
        if res.returncode != 0 or not lines:
            return {'error': 'tools/bench_tiled_one_gpu.py exited with %d' % res.returncode, 'stderr_tail': res.stderr.decode(errors='replace')[-1500:]}
        leg = json.loads(lines[-1])
        leg['leg_seconds'] = time.perf_counter() - t0
        leg['note'] = 'configs[4] asks for 8 GPUs; this is the same job with its eight ranks resident on ONE GPU (in-process transport): unmeasured on a multi-GPU node'
        return leg
    def one_engine_8192():
        # the same 8192 x 8192 job in ONE engine (tensors of 4 GiB and more: the BIG Winograd builds, 64-bit-addressed style gradient):
        # the independent reference of the sharded run (tests/test_gpu_parity.py) and what one MI355X does with the image un-sharded
        import ctypes
        free = ctypes.c_size_t()
        total = ctypes.c_size_t()
        hip = ctypes.CDLL('libamdhip64.so')
        rc = hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
        if rc != 0:
            return {'skipped': 'hipMemGetInfo failed (%d): the free HBM is unknown, the 175 GB job is not started' % rc}
        if free.value < 200 * 2 ** 30:
            return {'skipped': 'only %.0f GiB of HBM free, the un-sharded 8192 x 8192 job needs 175' % (free.value / 2 ** 30)}
        rs = np.random.RandomState
        content = rs(1).randint(0, 256, (8192, 8192, 3)).astype(np.uint8)
        init = rs(3).randint(0, 256, (8192, 8192, 3)).astype(np.uint8)
        style = rs(2).randint(0, 256, (1024, 1024, 3)).astype(np.uint8)
        return extra_config_leg('configs[4] image un-sharded: 8192x8192 in ONE engine (style image 1024x1024), Adam step 10, fp32',
                                (content, style, init, WEIGHTS, PARAMS), 'adam', 'fp32', 3, 2, 2, device)
    if os.environ.get('ST2_BENCH_SKIP_8192') != '1':
        guarded('8192x8192 one engine adam fp32', one_engine_8192)
        guarded('configs[4] 8192x8192 2x4 tiles, eight ranks on one GPU', tiled_8192)
    out['_seconds'] = time.perf_counter() - t_all
    return out


# ------------------------------------------------------------------------------------------------ multi-rank plumbing
def fan_out(args, argv):
    """--gpus N > 1 without a launcher: start N fresh processes, one per GPU, before anything here touches the GPU.
    Rank 0's stdout (the one JSON line) is relayed.  Every child is watched: when any rank exits non-zero (device out of
    range, out of memory, a failed rendezvous) the others are terminated at once and that code is returned, instead of
    rank 0 sitting in a barrier until the collective's own timeout."""
    import tempfile
    n = args.gpus
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs, out_file = [], tempfile.TemporaryFile()
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('ST2_RENDEZVOUS_TIMEOUT_S', '120')        # a rank that never arrives fails the others soon, not in 30 min
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out_file if rank == 0 else subprocess.DEVNULL))
    codes = [None] * n
    bad = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and bad is None:
                    bad = (r, codes[r])
        if bad is not None:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            deadline = time.time() + 10
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    out_file.seek(0)
    sys.stdout.write(out_file.read().decode())
    sys.stdout.flush()
    if bad is not None:
        print('bench.py: rank %d exited with code %d; the other ranks were stopped (exit codes %s)' % (bad[0], bad[1], codes), file=sys.stderr)
        return abs(bad[1]) or 1
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=5, help='timed blocks of --steps steps (value = the median block)')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--optimizer', default='adam', choices=['adam', 'lbfgs'])
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="bf16 = BASELINE configs[2] 'bf16 features / fp32 Gram'; the headline metric is fp32")
    ap.add_argument('--conv-algo', type=int, default=1, choices=[0, 1, 2],
                    help='st_set_conv_algo: 1 (default, the headline arithmetic) Winograd on the fp32 matrix cores, 0 direct kernel only, '
                         '2 Winograd with split operands on the bf16 matrix cores (fp32 results; reported as a custom workload)')
    ap.add_argument('--examples', action='store_true',
                    help='BASELINE configs[0]: the example pair fitted to 256 px by jobs.resize_to_fit, noise init, 50 Adam iterations, CPU oracle beside it')
    ap.add_argument('--examples-iters', type=int, default=EXAMPLES_ITERS, help=argparse.SUPPRESS)       # (the GPU test shortens the job)
    ap.add_argument('--tiled', default='', help='RxC: ONE image of --size tile-sharded over R*C GPUs (BASELINE configs[4])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-worker-level', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the short legs of BASELINE configs[2], [0] and [4]-on-one-GPU (default run, N = 1 only)')
    ap.add_argument('--cpu-size', type=int, default=0, help='image size of the CPU sample (default: --size)')
    ap.add_argument('--engine', default='hip', choices=['hip', 'stub'], help=argparse.SUPPRESS)
    ap.add_argument('--rehearse-one-gpu', action='store_true',
                    help='every rank uses device 0 and the ranks meet over gloo (RCCL refuses two ranks on one device): checks the '
                         '--gpus fan-out on a one-GPU box; the ranks share the card, so the rate is NOT a measurement')
    args = ap.parse_args(argv)
    if args.gpus < 1 or args.steps < 1 or args.repeats < 1:
        ap.error('--gpus, --steps and --repeats must be positive')

    # ---- ranks: from the launcher, or started here (nothing above this line has touched the GPU)
    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            return fan_out(args, argv)
    elif int(os.environ['WORLD_SIZE']) != args.gpus:
        print('bench.py: --gpus %d but the launcher started %s ranks' % (args.gpus, os.environ['WORLD_SIZE']), file=sys.stderr)
        return 2
    if args.tiled and args.gpus == 1 and args.tiled != '1x1':
        # one GPU, a grid of several ranks: every rank an engine context of this process, time-sliced (tools/bench_tiled_one_gpu.py)
        import runpy
        sys.argv = [os.path.join(HERE, 'tools', 'bench_tiled_one_gpu.py'), '--size', str(args.size), '--grid', args.tiled,
                    '--steps', str(args.steps), '--warmup', str(args.warmup), '--precision', 'bf16' if args.precision == 'bf16' else 'fp32',
                    '--optimizer', args.optimizer]
        runpy.run_path(sys.argv[0], run_name='__main__')
        return 0
    if args.tiled:
        import runpy
        sys.argv = [os.path.join(HERE, 'tools', 'bench_tiled.py'), '--size', str(args.size), '--grid', args.tiled,
                    '--steps', str(args.steps), '--warmup', str(args.warmup), '--optimizer', args.optimizer,
                    '--precision', args.precision]
        runpy.run_path(sys.argv[0], run_name='__main__')
        return 0

    if os.environ.get('ST2_BENCH_TEST_FAIL_RANK') == os.environ.get('RANK', '0') and args.engine == 'stub':
        return 7                    # tests/test_bench_fanout_cpu.py: a rank that dies before the rendezvous
    # native libraries (gloo, RCCL, the HIP runtime) print to fd 1 at will: keep the real stdout for the ONE JSON line
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    from style_transfer2_amd import distributed as st2_dist
    group = st2_dist.Group(backend='gloo' if (args.engine == 'stub' or args.rehearse_one_gpu) else None)
    rank, local_rank, world = group.rank, group.local_rank, group.world
    if args.rehearse_one_gpu:
        local_rank = 0

    inputs = None
    if args.engine == 'stub':
        job, sync = StubJob(), StubJob().sync
    else:
        if args.examples:
            args.size, args.optimizer = EXAMPLES_FIT, 'adam'
        inputs = job_inputs(args)
        job = make_job(inputs, args.optimizer, local_rank, args.precision, args.conv_algo)
        sync = job.engine.sync
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.engine == 'hip'
    cpu_size = args.cpu_size or args.size
    cpu_iters = args.examples_iters if args.examples else 2          # (two oracle iterations, timed one by one: a spread)
    dev_steps = cpu_iters + (1 if (args.optimizer == 'lbfgs' and cpu_iters == 1) else 0)      # (the oracle leg's untimed first L-BFGS step)
    dev_eval = device_eval_for_parity(job, dev_steps) if want_cpu and cpu_size == args.size else None

    blocks = []
    for r in range(args.repeats):
        blocks.append(st2_dist.timed_region(group, job.step_async, args.steps, args.warmup if r == 0 else 0, sync))
    elapsed = statistics.median(blocks)

    prof, prof_steps = {}, 0
    if args.engine == 'hip':
        # per-kernel-class HIP-event timing of the same steps (separate leg so `value` carries no event overhead)
        prof_steps = max(3, min(10, args.steps))
        prof = profiled_leg(job, prof_steps)

    if rank == 0:
        its = world * args.steps / elapsed
        f32 = args.precision == 'fp32'
        out = {
            'metric': 'style-transfer iters/sec @%dpx VGG19' % args.size,
            'value': its, 'unit': 'it/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': SPLIT_DTYPE if (args.conv_algo == 2 and f32) else 'f32' if f32 else 'bf16 conv operands, f32 accumulate/Gram/optimizer', 'data': 'synthetic',
            'images_per_hour': its * 3600.0 / ITERS_PER_IMAGE,
            'timing': {'blocks': args.repeats, 'steps_per_block': args.steps, 'value_from': 'median block',
                       'ms_per_step': {'median': 1e3 * elapsed / args.steps, 'min': 1e3 * min(blocks) / args.steps,
                                       'max': 1e3 * max(blocks) / args.steps},
                       'block_ms': [round(1e3 * b, 3) for b in blocks]},
            'config': {'workload': workload_label(args), 'jobs': world,
                       'parallelism': 'independent jobs, 1 per GPU, no collective (replicas)',
                       'scaling_curve': 'measured only by the driver (N = 1, 2, 4, 8); none has been measured by the builder'},
        }
        if args.rehearse_one_gpu:
            out['config']['rehearsal'] = 'all %d ranks shared ONE GPU (gloo rendezvous): plumbing check, the rate is not a measurement' % world
        if prof:
            # the dominant kernel class = every conv3x3 launch on the matrix cores (direct implicit GEMM + Winograd F(2x2,3x3)).
            # The engine records ALGORITHMIC flops per launch (direct convolution: 2*9*Cin*Cout*H*W); a Winograd
            # launch executes 4/9 of them on the MFMA pipe.
            peak = PEAK_F32_MFMA_TFLOPS if f32 else PEAK_BF16_MFMA_TFLOPS
            direct, wino, executed, algorithmic, ms, launches = conv_class(prof, f32)
            flops = sum(c['flops'] for c in direct + wino)
            sec = ms * 1e-3
            total_ms = sum(v['ms'] for v in prof.values())
            traffic, traffic_src = committed_pmc('pmc_traffic.json', args)
            mfma, mfma_src = committed_pmc('pmc_mfma.json', args)
            out['roofline'] = {
                'bound': 'mfma',
                'kernel': 'conv3x3 on the %s matrix cores (forward + dgrad launches; %d of %d launches Winograd F(2x2,3x3))'
                          % ('f32' if f32 else 'bf16', sum(c['launches'] for c in wino), launches),
                'achieved': executed, 'peak': peak, 'unit': 'TFLOP/s', 'frac': executed / peak,
                'achieved_is': 'FLOPs the MFMA pipe executes / kernel time (HIP events on the engine stream): direct launches '
                               '2*9*K*M*H*W, Winograd launches 4/9 of that',
                'algorithmic': algorithmic, 'algorithmic_over_peak': algorithmic / peak,
                'algorithmic_is': 'direct-convolution FLOPs (SURVEY 8d) / kernel time; exceeds the peak where Winograd runs '
                                  'because it needs 2.25x fewer multiplies -- not a fraction of any ceiling',
                'traffic': traffic['_conv3x3_all']['hbm_bytes_per_launch'] if traffic else None, 'traffic_source': traffic_src,
                'mfma_busy_pmc': ({k: round(v['mfma_busy_frac'], 3) for k, v in mfma.items() if k.startswith('conv3x3')} if mfma else None),
                'mfma_busy_source': mfma_src,
                'executed_flops_per_launch': executed * 1e12 * sec / launches if launches else 0.0,
                'algorithmic_flops_per_launch': flops / launches if launches else 0.0,
                'avg_launch_ms': ms / launches if launches else 0.0,
                'share_of_step': ms / total_ms if total_ms else 0.0}
            out['kernel_ms_per_step'] = {k: round(v['ms'] / prof_steps, 4) for k, v in sorted(prof.items())}
            # every kernel class against its own roofline (flops / bytes recorded by the engine per launch): matrix-core
            # classes in executed TFLOP/s vs the MFMA peak of the operand type, streaming passes in GB/s vs HBM 8 TB/s
            out['kernel_rooflines'] = {k: class_roofline(k, v, peak) for k, v in sorted(prof.items()) if v['ms'] > 0}
            split = split_conv_class(prof)
            if split:                    # --conv-algo 2: the split kernel's launches against the bf16 peak; the line above is what stayed on the fp32 pipe
                out['roofline_f32_pipe_rest'] = out['roofline']
                out['roofline'] = split
        if world == 1 and args.engine == 'hip' and not args.no_worker_level:
            out['worker_level'] = worker_level(job, max(3, min(20, args.steps)))
        if want_cpu:
            cpu_inputs = inputs if cpu_size == args.size else images(cpu_size) + (WEIGHTS, PARAMS)
            out['cpu_baseline'], parity = cpu_baseline_and_parity(cpu_inputs, args.optimizer, args.precision, dev_eval, cpu_iters)
            out['parity'] = parity
            if args.examples:
                out['config']['resize_to_fit_matches_reference_fixture'] = example_inputs()[3]
        default_headline = (args.size, args.optimizer, args.precision, args.conv_algo) == (1024, 'adam', 'fp32', 1) and not args.examples
        if world == 1 and args.engine == 'hip' and default_headline and not args.no_extra_configs:
            job.engine.close()                   # the headline job's HBM goes back before the other configs run
            out['extra_configs'] = extra_configs(args, local_rank)
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    group.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
