#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary (dusp_render_host = what the N-API addon calls): upload parameters,
render, download the PCM — into a result buffer from the context's pinned pool (what the addon and runtime.Program.render
hand out: one DMA) and into pageable memory (worker threads double-buffering pinned staging tiles)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dusp_amd as d  # noqa: E402
from dusp_amd import descriptor, runtime  # noqa: E402

sr = 48000
d.configure(sr)
ctx = runtime.Context(0, sr)
for V, seconds in [(1, 10), (64, 10), (1024, 1), (1024, 4)]:
    n = sr * seconds
    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Osc(10 * k), d.Ramp(n, 1, 0).trigger())) for k in (1, 2)])
    params = (10.0 * np.arange(1, V + 1)).astype(np.float32).reshape(1, V)
    prog = ctx.build(uni.words)
    for mode, pinned in (("pinned result buffer", True), ("pageable result buffer", False)):
        pcm = prog.render(n, V, params, pinned=pinned)
        pcm2 = prog.render(n, V, params, pinned=pinned)  # (two buffers alive, like a caller that still holds the last result)
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            del pcm
            pcm = prog.render(n, V, params, pinned=pinned)
            pcm, pcm2 = pcm2, pcm
        dt = (time.perf_counter() - t0) / reps
        print("host path, %-22s: %5d voices x %2d s: %8.2f ms per call  %8.1f Msamples/s  %6.2f GB/s of PCM (kernel %.3f ms)"
              % (mode, V, seconds, dt * 1e3, V * n / dt / 1e6, 4.0 * V * n / dt / 1e9, prog.last_kernel_ms()), flush=True)
        del pcm, pcm2
    prog.close()
