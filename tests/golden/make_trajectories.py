#!/usr/bin/env python3
"""Generates tests/golden/oracle_trajectories.npz: trajectories of the CPU ORACLE (oracle/, not the reference) that the GPU tests
compare the engine with -- stored so that the GPU suite does not spend a minute of box time re-running numpy, and re-checked on the
CPU by tests/test_oracle_golden.py::test_stored_oracle_trajectories_are_what_the_oracle_computes (first steps of each).

  drift_*    the reference's example pair fitted to 256 px (192 x 256), iterate-like initial image, initial_weights.yaml losses,
             11 L-BFGS steps with step 1: oracle in fp32 and the same oracle with every conv operand rounded to bf16
             (tests/test_gpu_fullsize.py::test_bf16_engine_drifts_from_fp32_no_more_than_the_rounded_operand_oracle_does)
  config1_*  BASELINE configs[0]: tests/golden/config1_inputs.npz, noise initial image, Adam step 10, all 50 iterations
             (tests/test_gpu_parity.py::test_config1_golden_gate_starry_night_256px_adam_iters)
  size_*     the same pair fitted to 1024 px (content 768 x 1024), iterate-like initial image, L-BFGS step 1 (round 5): the per-step
             losses of the fp32 oracle over five steps and of the ROUNDED-OPERAND oracle (bf16 conv operands) over three, scalars
             only, plus every fourth pixel (both directions) of the fp32 oracle's fifth iterate: the image MSE the test bounds (bar: 5 % of
             a move of ~900, measured 0.13) is taken over that sample
             (tests/test_gpu_fullsize.py::test_image_like_job_*_follows_the_*oracle*)

Run here: python tests/golden/make_trajectories.py [drift] [config1] [size]   (no argument: all three; a part that is not named
keeps its stored arrays; all of it is about ten minutes on 8 cores)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle                                            # noqa: E402

WEIGHTS = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1, 'conv5_1': 1}, 'deepdream': {}}
PARAMS = {'p': 50, 'p_power': 6, 'tv': 5, 'tv_power': 2}
WEIGHTED = ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv4_2', 'conv5_1']
CONFIG1_WEIGHTS = {'content': {'conv4_2': 0.08}, 'style': {'conv1_1': 1, 'conv2_1': 1, 'conv3_1': 1, 'conv4_1': 1}, 'deepdream': {}}


def image_like(fit):
    from PIL import Image
    from style_transfer2_amd import jobs
    src = np.load(os.path.join(HERE, 'config1_sources.npz'))
    content = np.uint8(jobs.resize_to_fit(Image.fromarray(src['golden_gate']), fit))
    style = np.uint8(jobs.resize_to_fit(Image.fromarray(src['starry_night']), fit))
    init = np.clip(content.astype(np.int32) + np.random.RandomState(5).randint(-16, 17, content.shape), 0, 255).astype(np.uint8)
    return content, style, init


def drift_run(operands, steps, fit=256):
    content, style, init = image_like(fit)
    topo = oracle.VGG19_TOPOLOGY
    job = oracle.TransferOracle(oracle.NetOracle(topo, oracle.he_init_weights(topo, seed=0), full_forward=False, operands=operands))
    job.feature_layers = WEIGHTED
    job.set_input(init); job.set_content(content); job.set_style(style); job.reset()
    job.set_weights(WEIGHTS, PARAMS)
    job.set_optimizer('lbfgs', 1)
    assert job.start()
    losses = []
    img = None
    for _ in range(steps):
        img, tr = job.step()
        losses.append(tr['loss'])
    return np.array(losses, np.float64), np.asarray(img, np.float32)


def config1_run(steps):
    g = np.load(os.path.join(HERE, 'config1_inputs.npz'))
    content, style = g['golden_gate'], g['starry_night']
    init = np.random.RandomState(3).randint(0, 256, content.shape).astype(np.uint8)
    topo = oracle.VGG19_TOPOLOGY
    cpu = oracle.TransferOracle(oracle.NetOracle(topo, oracle.he_init_weights(topo, seed=0), full_forward=False))
    cpu.set_input(init); cpu.set_content(content); cpu.set_style(style); cpu.reset()
    cpu.set_weights(CONFIG1_WEIGHTS, PARAMS)
    cpu.set_optimizer('adam', 10)
    assert cpu.start()
    losses, first, img = [], None, None
    for i in range(steps):
        img, tr = cpu.step()
        losses.append(tr['loss'])
        if i == 0:
            first = tr
    keys = [str(k) for k in first]
    return np.array(losses, np.float64), np.asarray(img, np.float32), keys, np.array([float(first[k]) for k in first], np.float64)


if __name__ == '__main__':
    path = os.path.join(HERE, 'oracle_trajectories.npz')
    parts = set(sys.argv[1:]) or {'drift', 'config1', 'size'}
    data = dict(np.load(path)) if os.path.exists(path) else {}
    if 'drift' in parts:
        lo32, io32 = drift_run('fp32', 11)
        lo16, io16 = drift_run('bf16', 11)
        data.update(drift_losses_fp32=lo32, drift_losses_bf16=lo16, drift_final_fp32=io32, drift_final_bf16=io16)
        print('drift fp32', lo32, '\nbf16', lo16)
    if 'config1' in parts:
        lc, ic, keys, vals = config1_run(50)
        data.update(config1_losses=lc, config1_final=ic, config1_first_keys=np.array(keys), config1_first_values=vals)
        print('config1', lc)
    if 'size' in parts:
        ls32, is32 = drift_run('fp32', 5, fit=1024)
        ls16, _ = drift_run('bf16', 3, fit=1024)
        data.pop('size_final_fp32_u8', None); data.pop('size_final_fp32_clipped_frac', None)
        data.update(size_losses_fp32=ls32, size_losses_bf16=ls16, size_final_fp32_sub4=np.ascontiguousarray(is32[::4, ::4, :]))
        print('size fp32', ls32, '\nbf16', ls16)
    np.savez_compressed(path, **data)
