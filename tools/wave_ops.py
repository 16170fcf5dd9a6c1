#!/usr/bin/env python3
"""Per-unit cost on the wave engine: small graphs forced onto ENGINE_WAVE, 16384 instances x 1 s (time split off).
  python tools/wave_ops.py ["--only=name;name"]"""
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
    "osc(k)": lambda k: d.Osc(110 + k / 8),
    "osc(k) x 4": lambda k: d.Sum(d.Sum(d.Osc(110 + k / 8), d.Osc(50 + k / 16)), d.Sum(d.Osc(70 + k / 4), d.Osc(30 + k / 2))),
    "fm: osc(osc*40+220)": lambda k: d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + k / 100), 40), 220 + k / 4)),
    "ramp": lambda k: d.Multiply(d.Ramp(sr, 1, 0).trigger(), 0.5 + k / 1e5),
    "mul(osc, 0.5)": lambda k: d.Multiply(d.Osc(110 + k / 8), 0.5),
    "shape(k)": lambda k: d.Shape("decay", 0.5 + k / 1e4).trigger(),
    "mul(osc, shape(k))": lambda k: d.Multiply(d.Osc(110 + k / 8), d.Shape("decay", 0.5 + k / 1e4).trigger()),
    "timer * k": lambda k: d.Multiply(d.Timer(), 1.0 + k / 1e4),
    "filter(osc)": lambda k: d.Filter(d.Osc(110 + k / 8), 800),
    "allpass(osc)": lambda k: (lambda a: (setattr(a, "IN", d.Osc(110 + k / 8)), a)[1])(d.AllPass(0.0021, 0.6)),
    "filter(osc) * ramp": lambda k: d.Multiply(d.Filter(d.Osc(110 + k / 8), 800 + k / 16), d.Ramp(sr, 1, 0).trigger()),
    "filter(osc, lfo)": lambda k: d.Filter(d.Osc(110 + k / 8), d.Sum(d.Multiply(d.Osc(5), 800), 1000)),
    "delay(osc, 300.5)": lambda k: d.Delay(d.Osc(110 + k / 8), 300.5, 2048),
    "delay(osc, 30.5)": lambda k: d.Delay(d.Osc(110 + k / 8), 30.5, 2048),
    "delay(osc, lfo)": lambda k: d.Delay(d.Osc(110 + k / 8), d.Sum(d.Multiply(d.Osc(2), 40), 200), 1024),
    "comb(osc)": lambda k: (lambda a: (setattr(a, "IN", d.Osc(110 + k / 8)), a)[1])(d.CombFilter(0.004, 0.7)),
    "osc * ahd": lambda k: d.Multiply(d.Osc(110 + k / 8), d.AHD(0.01, 0.1, 0.5).trigger()),
    "pan(osc, 0.25)": lambda k: d.Pan(d.Osc(110 + k / 8), 0.25),
    "multiosc(2 channels)": lambda k: d.MultiChannelOsc([220 + k / 8, 330]),
    "srr(osc, 5)": lambda k: d.SampleRateRedux(d.Osc(110 + k / 8), 5),
    "filter(filter(osc))": lambda k: d.Filter(d.Filter(d.Osc(110 + k / 8), 900), 1200),
}
only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
for name, g in graphs.items():
    if only and name not in only[0].split(";"):
        continue
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
    ms = float(np.median(ts))
    prog._read_info()
    print("%-22s %8.3f ms  %8.1f Gsamples/s   [%s]" % (name, ms, V * n / ms / 1e6, prog.shape), flush=True)
    prog.close()
    del out
