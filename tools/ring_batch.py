#!/usr/bin/env python3
"""Delay lines that go through ordered slot operations (wave engine) vs the chunk engine, by batch size:
   python tools/ring_batch.py        (1 s per instance)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import dusp_amd as d  # noqa: E402
from dusp_amd import descriptor, runtime  # noqa: E402

sr = 48000
d.configure(sr)
ctx = runtime.Context(0, sr)
stream = torch.cuda.current_stream().cuda_stream


def seconds_to_samples(x):
    s = d.SecondsToSamples()
    s.IN = x
    return s


graphs = {
    "delay, signal-rate (SimpleDelay core)": lambda k: d.Delay(d.Osc(110 + k), seconds_to_samples(d.Sum(d.Multiply(d.Osc(0.5), 0.001), 0.25)), sr),
    "delay, 30.5 samples": lambda k: d.Delay(d.Osc(110 + k), 30.5, 4096),
    "monodelay, signal-rate (Space core)": lambda k: d.MonoDelay(d.Osc(110 + k), d.Sum(d.Multiply(d.Osc(0.5), 100), 400)),
}
n = sr
for name, g in graphs.items():
    full = descriptor.unify([descriptor.extract(g(k)) for k in (0, 8, 16)])
    base = full.params[:, 0].astype(np.float64)
    step = (full.params[:, 1].astype(np.float64) - base) / 8.0
    for V in (1, 64, 1024, 8192):
        params = (base[:, None] + step[:, None] * np.arange(V)[None, :]).astype(np.float32)
        dp = torch.from_numpy(np.ascontiguousarray(params)).cuda()
        row = []
        for engine in (runtime.ENGINE_WAVE, runtime.ENGINE_CHUNK):
            prog = ctx.build(full.words, engine)
            out = torch.empty((V, prog.n_out_channels, n), dtype=torch.float32, device="cuda")
            ts = []
            for r in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), stream)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            row.append(float(np.median(ts)))
            prog.close()
            del out
        print("%-40s inst=%-5d wave %9.3f ms   chunk %9.3f ms   (%.1fx)" % (name, V, row[0], row[1], row[1] / row[0]), flush=True)
