#!/usr/bin/env python3
"""Worker-level rate: one Iterate (HxWx3 float32, D2H over PCIe + pickle, as pyzmq's send_pyobj does) per step,
with and without the asynchronous sender thread.  The device-resident rate (bench.py `value`) excludes this."""
import os, pickle, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, messages, worker as worker_mod

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30


class PickleSink:                       # what send_pyobj costs on the sending side, minus the socket write
    def __init__(self):
        self.bytes = 0

    def send_pyobj(self, obj):
        self.bytes += len(pickle.dumps(obj, protocol=pickle.DEFAULT_PROTOCOL))


job = bench.make_job(bench.images(size) + (bench.WEIGHTS, bench.PARAMS), 'adam', 0)
for mode in ('device-resident (no Iterate)', 'synchronous Iterate', 'asynchronous Iterate'):
    for _ in range(3):
        job.step_async()
    job.engine.sync()
    sink = PickleSink()
    out = worker_mod.AsyncSender(sink) if mode.startswith('async') else sink
    t0 = time.perf_counter()
    for _ in range(steps):
        if mode.startswith('device'):
            job.step_async()
        else:
            image, trace = job.step()
            out.send_pyobj(messages.Iterate(image, job.t, trace))
    job.engine.sync()
    if mode.startswith('async'):
        out.close()
    dt = time.perf_counter() - t0
    print('%-30s %6.2f it/s  (%.2f ms/step, %.1f MB pickled per step)' % (mode, steps / dt, 1e3 * dt / steps, sink.bytes / steps / 1e6), flush=True)
