#!/usr/bin/env python3
"""Wave engine vs chunk engine vs what AUTO picks, by batch size, for a few general graphs: python tools/engine_batch.py   (1 s per instance)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DUSP_WAVE_SEGMENTS", "1")
import torch  # noqa: E402
import dusp_amd as d  # noqa: E402
from dusp_amd import descriptor, runtime  # noqa: E402

sr = 48000
d.configure(sr)
ctx = runtime.Context(0, sr)
stream = torch.cuda.current_stream().cuda_stream


def allpass(k):
    a = d.AllPass(0.0021, 0.6)
    a.IN = d.Osc(110 + k / 8)
    return a


graphs = {
    "osc(k) x 4": lambda k: d.Sum(d.Sum(d.Osc(110 + k / 8), d.Osc(50 + k / 16)), d.Sum(d.Osc(70 + k / 4), d.Osc(30 + k / 2))),
    "fm: osc(osc*40+220)": lambda k: d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + k / 100), 40), 220 + k / 4)),
    "filter(osc)": lambda k: d.Filter(d.Osc(110 + k / 8), 800),
    "allpass(osc)": allpass,
    "env voice: osc * shape": lambda k: d.Multiply(d.Osc(110 + k / 8), d.Shape("decay", 0.5).trigger()),
    "delay(osc, 300.5)": lambda k: d.Delay(d.Osc(110 + k / 8), 300.5, 4096),
}
n = sr
for name, g in graphs.items():
    full = descriptor.unify([descriptor.extract(g(k)) for k in (0, 8, 16)])
    base = full.params[:, 0].astype(np.float64)
    step = (full.params[:, 1].astype(np.float64) - base) / 8.0
    for V in (256, 4096, 32768):
        params = (base[:, None] + step[:, None] * np.arange(V)[None, :]).astype(np.float32)
        dp = torch.from_numpy(np.ascontiguousarray(params)).cuda()
        row = []
        auto_name = ''
        for engine in (runtime.ENGINE_WAVE, runtime.ENGINE_CHUNK, runtime.ENGINE_AUTO):
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
            auto_name = prog.engine
            prog.close()
            del out
        print("%-26s inst=%-6d wave %9.3f ms   chunk %9.3f ms   auto(%s) %9.3f ms" % (name, V, row[0], row[1], auto_name, row[2]), flush=True)
