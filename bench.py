#!/usr/bin/env python3
"""Benchmark of the style-transfer inner loop on MI355X (BASELINE.json: iterations/sec @1024px VGG19).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 1024] [--optimizer adam]

One "step" = one ``StyleTransfer.step()`` = forward to conv5_1, 1 content + 5 style loss terms,
ranged backward, fused TV/p-norm/Adam pass (reference worker.py:303-310).  Inputs are synthetic
(seeded He-normal VGG19 weights, uniform-noise uint8 images: SURVEY section 8d) and resident in HBM
before the timed region; nothing is read back inside it.  With N > 1 (launched by torch.distributed.run,
one rank per GPU) every rank runs an independent job -- jobs are the unit the reference scales by
(one worker per GPU, no collective) -- so `value` is the aggregate it/s and scaling is weak.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the conv3x3 MFMA implicit GEMM,
forward + data-gradient launches) from HIP events on the engine's own stream; `cpu_baseline` is the
CPU oracle (kind "port") timed on this host on a bounded sample, N = 1 only.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import style_transfer2_amd as st2                      # noqa: E402
from style_transfer2_amd import weights as st2_weights  # noqa: E402
from style_transfer2_amd import distributed as st2_dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E
PEAK_BF16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (8 passes x 4 cyc)
WEIGHTS = {'content': {'conv4_2': 0.08},
           'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1},
           'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
ITERS_PER_IMAGE = 500


def images(size):
    rs = np.random.RandomState
    shape = (size, size, 3)
    return (rs(1).randint(0, 256, shape).astype(np.uint8), rs(2).randint(0, 256, shape).astype(np.uint8),
            rs(3).randint(0, 256, shape).astype(np.uint8))


def make_job(size, optimizer, device, precision='fp32'):
    content, style, init = images(size)
    model = st2.HipModel(st2_weights.he_normal(st2.VGG19_TOPOLOGY, seed=0), device=device, precision=precision)
    job = st2.StyleTransfer(model)
    job.set_input(init)
    job.set_content(content)
    job.set_style(style)
    job.set_weights(WEIGHTS, PARAMS)
    job.optimizer_cls = {'adam': st2.AdamOptimizer, 'lbfgs': st2.LBFGSOptimizer}[optimizer]
    job.set_step_size({'adam': 10, 'lbfgs': 1}[optimizer])
    job.reset()
    assert job.start()
    return job


def class_roofline(name, rec, conv_peak):
    """One kernel class against its rooflines: TFLOP/s vs the MFMA peak of its operand type when the engine recorded
    algorithmic flops for it, GB/s vs HBM when it recorded algorithmic bytes (both for classes that have both)."""
    sec = rec['ms'] * 1e-3
    out = {}
    if rec['flops'] > 0:
        tf = rec['flops'] / sec / 1e12
        out['TFLOP/s'] = round(tf, 1)
        out['frac_mfma'] = round(tf / (conv_peak if name.startswith('conv3x3') else PEAK_F32_MFMA_TFLOPS), 3)
    if rec['bytes'] > 0:
        gbs = rec['bytes'] / sec / 1e9
        out['GB/s'] = round(gbs, 1)
        out['frac_hbm'] = round(gbs / PEAK_HBM_GBS, 3)
    return out


def cpu_baseline(size, optimizer):
    """The CPU oracle ("port") on this host: setup, one untimed step (norm capture), one timed step."""
    import oracle
    from threadpoolctl import threadpool_info
    content, style, init = images(size)
    topo = oracle.VGG19_TOPOLOGY
    job = oracle.TransferOracle(oracle.NetOracle(topo, oracle.he_init_weights(topo, seed=0), full_forward=False))
    job.set_input(init)
    job.set_content(content)
    job.set_style(style)
    job.reset()
    job.set_weights(WEIGHTS, PARAMS)
    job.set_optimizer(optimizer, {'adam': 10, 'lbfgs': 1}[optimizer])
    job.start()
    job.step()
    t0 = time.perf_counter()
    job.step()
    dt = time.perf_counter() - t0
    threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    return {'value': 1.0 / dt, 'unit': 'it/s', 'cores': threads, 'host_cpus': os.cpu_count(), 'kind': 'port',
            'sample': '1 %s iteration at %dx%d (after 1 untimed), numpy+OpenBLAS oracle, forward stops at conv5_1'
                      % (optimizer, size, size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--optimizer', default='adam', choices=['adam', 'lbfgs'])
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="bf16 = BASELINE configs[3] 'bf16 features / fp32 Gram'; the headline metric is fp32")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-size', type=int, default=0, help='image size of the CPU sample (default: --size)')
    args = ap.parse_args()

    group = st2_dist.Group()
    rank, local_rank, world = group.rank, group.local_rank, group.world

    job = make_job(args.size, args.optimizer, local_rank, args.precision)
    elapsed = st2_dist.timed_region(group, job.step_async, args.steps, args.warmup, job.engine.sync)

    # per-kernel-class HIP-event timing of the same steps (separate leg so `value` carries no event overhead)
    prof_steps = max(3, min(10, args.steps))
    job.engine.profile_enable(True)
    for _ in range(prof_steps):
        job.step_async()
    prof = job.engine.profile_read()
    job.engine.profile_enable(False)

    if rank == 0:
        its = world * args.steps / elapsed
        # the dominant kernel = every conv3x3 launch on the matrix cores (direct implicit GEMM and Winograd F(2x2,3x3));
        # `flops` are ALGORITHMIC (direct-convolution: 2*9*Cin*Cout*H*W per launch) for both; the Winograd
        # launches execute 4/9 of them on the MFMA pipe, which `executed` accounts for.
        direct = [prof[k] for k in ('conv3x3_fwd_mfma_f32', 'conv3x3_dgrad_mfma_f32') if k in prof]
        wino = [prof[k] for k in ('conv3x3_fwd_wino_f32', 'conv3x3_dgrad_wino_f32') if k in prof]
        conv = direct + wino
        flops = sum(c['flops'] for c in conv)
        ms = sum(c['ms'] for c in conv)
        launches = sum(c['launches'] for c in conv)
        achieved = flops / (ms * 1e-3) / 1e12 if ms else 0.0
        executed = (sum(c['flops'] for c in direct) + sum(c['flops'] for c in wino) * 4.0 / 9.0) / (ms * 1e-3) / 1e12 if ms else 0.0
        total_ms = sum(v['ms'] for v in prof.values())
        # HBM bytes per conv launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE doubled for wide loads, WRITE_SIZE exact: tools/pmc_traffic.py); null when absent.
        traffic = None
        tpath = os.path.join(HERE, 'profiles', 'pmc_traffic.json')
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == 'fp32' else PEAK_BF16_MFMA_TFLOPS
        if args.size == 1024 and args.precision == 'fp32' and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))['_conv3x3_all']['hbm_bytes_per_launch']
            except (KeyError, ValueError):
                traffic = None
        # matrix-pipe busy fraction of the conv kernels from the committed rocprofv3 PMC pass of this same command
        # (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): tools/pmc_mfma.py); null when absent
        mfma_busy = None
        mpath = os.path.join(HERE, 'profiles', 'pmc_mfma.json')
        if args.size == 1024 and args.precision == 'fp32' and os.path.exists(mpath):
            try:
                mfma_busy = {k: round(v['mfma_busy_frac'], 3) for k, v in json.load(open(mpath)).items() if k.startswith('conv3x3')}
            except (KeyError, ValueError):
                mfma_busy = None
        out = {
            'metric': 'style-transfer iters/sec @%dpx VGG19' % args.size,
            'value': its, 'unit': 'it/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'bf16 conv operands, f32 accumulate/Gram/optimizer', 'data': 'synthetic',
            'images_per_hour': its * 3600.0 / ITERS_PER_IMAGE,
            'config': {'workload': 'configs[1]: %dx%d single image per GPU, VGG19 to conv5_1, content conv4_2 + 5 style '
                                   'layers, %s %s, %d iterations per image' % (args.size, args.size, args.optimizer,
                                                                                args.precision, ITERS_PER_IMAGE),
                       'jobs': world, 'parallelism': 'independent jobs, 1 per GPU, no collective'},
            'roofline': {'bound': 'mfma', 'kernel': 'conv3x3 on the %s matrix cores (forward + dgrad launches; %d of %d launches Winograd F(2x2,3x3))'
                                   % ('f32' if args.precision == 'fp32' else 'bf16', sum(c['launches'] for c in wino), launches),
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic,
                         'executed': executed, 'executed_frac': executed / peak, 'mfma_busy_pmc': mfma_busy,
                         'note': 'achieved = algorithmic direct-conv flops / kernel time, so frac can exceed 1 where Winograd '
                                 'runs; executed = flops the MFMA pipe actually performs (Winograd: 4/9 of algorithmic)',
                         'flops_per_launch': flops / launches if launches else 0.0,
                         'avg_launch_ms': ms / launches if launches else 0.0,
                         'share_of_step': ms / total_ms if total_ms else 0.0},
            'kernel_ms_per_step': {k: round(v['ms'] / prof_steps, 4) for k, v in sorted(prof.items())},
            # every kernel class against its own roofline (algorithmic flops / bytes recorded by the engine per launch):
            # matrix-core classes in TFLOP/s vs the MFMA peak of the operand type, streaming passes in GB/s vs HBM 8 TB/s
            'kernel_rooflines': {k: class_roofline(k, v, peak) for k, v in sorted(prof.items()) if v['ms'] > 0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.cpu_size or args.size, args.optimizer)
        print(json.dumps(out))
    group.close()


if __name__ == '__main__':
    main()
