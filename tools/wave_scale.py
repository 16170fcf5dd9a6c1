#!/usr/bin/env python3
"""Throughput of the general engines on multi-instance workloads that are NOT fused shapes:
FM voice (feed-forward), filtered oscillator, and the configs[3] feedback loop forced through each engine."""
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
ENG = {"auto": runtime.ENGINE_AUTO, "chunk": runtime.ENGINE_CHUNK, "wave": runtime.ENGINE_WAVE}


def fm(k):
    return d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + k / 100), 40), 220 + k / 4)), d.Ramp(sr, 1, 0).trigger())


def filt(k):
    return d.Filter(d.Multiply(d.Osc(110 + k / 8), d.Osc(2 + k / 1000)), 800 + k / 4)


def loop(k):
    s = d.Sum(d.Osc(110 + k / 64), 0)
    f = d.Filter(d.Delay(s, 480, 4096), 2000)
    s.B = d.Multiply(f, 0.5)
    return f


def run(name, graph, V, n, engines):
    uni = descriptor.unify([descriptor.extract(graph(k)) for k in (0, 8)])
    exs = [descriptor.extract(graph(k)) for k in range(0, V * 8, 8)] if V <= 64 else None
    # parameter columns follow the order of the differing constants in the descriptor
    cols = []
    full = descriptor.unify([descriptor.extract(graph(k)) for k in (0, 8, 16)])
    base = full.params[:, 0].astype(np.float64)
    step = (full.params[:, 1].astype(np.float64) - base) / 8.0
    params = (base[:, None] + step[:, None] * np.arange(V)[None, :]).astype(np.float32)
    dp = torch.from_numpy(np.ascontiguousarray(params)).cuda()
    for e in engines:
        try:
            prog = ctx.build(uni.words, ENG[e])
        except runtime.DuspHipError as err:
            print("%-14s %-6s  not applicable (%s)" % (name, e, err.message[:60]))
            continue
        out = torch.empty((V, prog.n_out_channels, n), dtype=torch.float32, device="cuda")
        ts = []
        for r in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), stream)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        print("%-14s %-6s -> %-5s %6d inst x %6d: %9.3f ms  %10.1f Msamples/s  %7.1f GB/s  [%s]"
              % (name, e, prog.engine, V, n, ms, V * n / ms / 1e3, 4.0 * V * n / ms / 1e6, prog.shape), flush=True)
        prog.close()
        del out


run("fm voice", fm, 16384, 48000, ["auto"])
run("filter voice", filt, 8192, 48000, ["auto", "chunk"])
run("feedback loop", loop, 8192, 48000, ["auto", "wave", "chunk"])
