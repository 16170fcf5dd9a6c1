#!/usr/bin/env python3
"""EXPERIMENT (off by default): DUSP_FILTER_FMA=1 — the Filter stage's recurrence with one fused multiply-add on its dependent chain
(fma(-b1, y1, P - b2 y2): three dependent instructions a step instead of five; ANOTHER rounding than Filter.js:40-46's two).
BASELINE configs[3] (8192 feedback loops x 10 s) both ways: kernel time, and the largest difference between the two renders over every
instance and sample, relative to full scale; three instances against the oracle as well.
  python tools/filter_fma_experiment.py [--seconds=10]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DUSP_WAVE_JIT", "2")
import torch  # noqa: E402
import dusp_amd as d  # noqa: E402
from dusp_amd import descriptor, runtime  # noqa: E402
from oracle import oracle  # noqa: E402  (the checker)

sr = 48000
d.configure(sr)
seconds = float(([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--seconds=")] or ["10"])[0])
n, V = int(seconds * sr), 8192


def loop(k):
    s = d.Sum(d.Osc(110 + k / 64), 0)
    f = d.Filter(d.Delay(s, 480, 4096), 2000)
    s.B = d.Multiply(f, 0.5)
    return f


uni = descriptor.unify([descriptor.extract(loop(k)) for k in (0, 64)])
params = (110 + np.arange(V) / 64.0).astype(np.float32).reshape(1, -1)
dp = torch.from_numpy(params).cuda()
outs = {}
for name, knob in (("as the reference rounds", None), ("DUSP_FILTER_FMA=1", "1")):
    if knob:
        os.environ["DUSP_FILTER_FMA"] = knob
    ctx = runtime.Context(0, sr)  # (knobs are read when a context is created)
    os.environ.pop("DUSP_FILTER_FMA", None)
    prog = ctx.build(uni.words, runtime.ENGINE_WAVE)
    out = torch.empty((V, 1, n), dtype=torch.float32, device="cuda")
    ts = []
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    prog._read_info()
    print("%-26s %8.3f ms   [%s]" % (name, float(np.median(ts[1:])), prog.shape), flush=True)
    outs[name] = out
    prog.close()
a, b = outs["as the reference rounds"], outs["DUSP_FILTER_FMA=1"]
scale = float(a.abs().max())
print("largest difference between the two renders: %.3g of full scale (%.3g), over %d x %d samples" % (float((a - b).abs().max()) / scale, scale, V, n))
for i in (0, 4097, 8191):
    want = oracle.render(uni.words, n, params=params, n_instances=V, instance=i)[0].astype(np.float64)
    for name, out in outs.items():
        got = out[i, 0].cpu().numpy().astype(np.float64)
        print("instance %4d against the oracle, %-26s max error %.3g of full scale" % (i, name, float(np.max(np.abs(got - want))) / max(1e-30, float(np.max(np.abs(want))))))
