#!/usr/bin/env python3
"""Big circuits: N voices summed by Sum.many where a voice is more than the fused sum chain takes (FM pairs, enveloped FM pairs, shaped
oscillators), 256 instances x 1 s.  Isomorphic voices of oscillators / Ramps / Multiply / Sum / maps run in a LOOP from 96 units on
(DUSP_JIT_LOOP_VOICES; =4096: never), other circuits as straight-line code up to DUSP_JIT_MAX_UNITS (256) and on the interpreter beyond.
  python tools/big_circuits.py [--voices=24,48,96] [--single]      (--single: ONE circuit of N voices, 10 s, cut in time over the chip)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--single" not in sys.argv:
    os.environ.setdefault("DUSP_WAVE_SEGMENTS", "1")
os.environ.setdefault("DUSP_WAVE_JIT", "2")
import torch  # noqa: E402
import dusp_amd as d  # noqa: E402
from dusp_amd import descriptor, runtime  # noqa: E402

sr = 48000
d.configure(sr)
ctx = runtime.Context(0, sr)
stream = torch.cuda.current_stream().cuda_stream
voices = [int(v) for v in ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--voices=")] or ["24,48,96"])[0].split(",")]
kinds = {
    "fm": lambda k, j: d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + j / 7 + k / 100), 40), 220 + 11.5 * j + k / 4)),
    "fm*ramp": lambda k, j: d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + j / 7 + k / 100), 40), 220 + 11.5 * j + k / 4)), d.Ramp(24000 + 100 * j, 1, 0).trigger()),
    "osc*shape": lambda k, j: d.Multiply(d.Osc(110 + 7.25 * j + k / 8), d.Shape("decay", 0.3 + j / 50).trigger()),
}
V, n = (1, 480000) if "--single" in sys.argv else (256, 48000)
for kind, voice in kinds.items():
    for nv in voices:
        full = descriptor.unify([descriptor.extract(d.Sum.many([voice(k, j) for j in range(nv)])) for k in (0, 8)])
        base = full.params[:, 0].astype(np.float64)
        step = (full.params[:, 1].astype(np.float64) - base) / 8.0
        params = (base[:, None] + step[:, None] * np.arange(V)[None, :]).astype(np.float32)
        dp = torch.from_numpy(np.ascontiguousarray(params)).cuda() if params.size else None
        t0 = time.perf_counter()
        prog = ctx.build(full.words, runtime.ENGINE_WAVE)
        out = torch.empty((V, prog.n_out_channels, n), dtype=torch.float32, device="cuda")
        ts = []
        for r in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t1 = time.perf_counter()
            a.record()
            prog.render_device(n, V, dp.data_ptr() if dp is not None else None, out.data_ptr(), stream)
            b.record()
            torch.cuda.synchronize()
            ts.append((a.elapsed_time(b), time.perf_counter() - t1))
        prog._read_info()
        print(("one circuit, 10 s: " if V == 1 else "") + "%-10s x %3d voices (%4d units)  first call %7.2f s   then %8.3f ms   [%s]  checksum %.6g" %
              (kind, nv, prog.n_units, ts[0][1], float(np.median([t[0] for t in ts[1:]])), prog.shape, float(out[0, 0, :4096].double().sum())), flush=True)
        prog.close()
        del out
