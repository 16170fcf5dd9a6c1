#!/usr/bin/env python3
"""Where a Filter voice's time goes on its compiled kernel: the same 16384 x 1 s render with cheaper and cheaper feed-forward parts.
  DUSP_WAVE_PER_WAVE=1|2 python tools/filter_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DUSP_WAVE_SEGMENTS", "1")
os.environ.setdefault("DUSP_WAVE_JIT", "2")  # wait for every circuit's compiled kernel (the default lets short renders run interpreted meanwhile)
import torch  # noqa: E402
import dusp_amd as d  # noqa: E402
from dusp_amd import descriptor, runtime  # noqa: E402

sr = 48000
d.configure(sr)
ctx = runtime.Context(0, sr)
stream = torch.cuda.current_stream().cuda_stream
V, n = 16384, 48000
graphs = {
    "filter(osc)": lambda k: d.Filter(d.Osc(110 + k / 8), 800),
    "filter(ramp*k)": lambda k: d.Filter(d.Multiply(d.Ramp(sr, 1, 0).trigger(), 0.5 + k / 1e5), 800),
    "filter(filter(ramp*k))": lambda k: d.Filter(d.Filter(d.Multiply(d.Ramp(sr, 1, 0).trigger(), 0.5 + k / 1e5), 800), 1200),
    "osc": lambda k: d.Osc(110 + k / 8),
}
for name, g in graphs.items():
    full = descriptor.unify([descriptor.extract(g(k)) for k in (0, 8, 16)])
    base = full.params[:, 0].astype(np.float64)
    step = (full.params[:, 1].astype(np.float64) - base) / 8.0
    params = (base[:, None] + step[:, None] * np.arange(V)[None, :]).astype(np.float32)
    dp = torch.from_numpy(np.ascontiguousarray(params)).cuda()
    prog = ctx.build(full.words, runtime.ENGINE_WAVE)
    out = torch.empty((V, prog.n_out_channels, n), dtype=torch.float32, device="cuda")
    ts = []
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), stream)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    prog._read_info()
    print("%-26s %8.3f ms   [%s]" % (name, float(np.median(ts)), prog.shape), flush=True)
    prog.close()
