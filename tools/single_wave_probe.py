#!/usr/bin/env python3
"""One circuit on the wave engine, 10 s: what a chunk costs.  Meant to run under rocprofv3 --pmc (instruction counters of the
single wavefront that does the work):  python tools/single_wave_probe.py [filter|delay|retrigger|osc]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DUSP_WAVE_SEGMENTS", "1")
import dusp_amd as d  # noqa: E402
from dusp_amd import descriptor, render, runtime  # noqa: E402

d.configure(48000)
which = sys.argv[1] if len(sys.argv) > 1 else "filter"
g = {"filter": lambda: d.Filter(d.Osc(110, "saw"), 800),
     "delay": lambda: d.Delay(d.Osc(500), 300.5, 4096),
     "osc": lambda: d.Osc(440.5),
     "shortdelay": lambda: d.Delay(d.Osc(500), 30.5, 4096)}[which]()
prog = render.context(48000).build(descriptor.extract(g).words, runtime.ENGINE_WAVE)
n = 480000
prog.render(n)
prog.render(n)
print(which, prog.shape, "kernel ms", prog.last_kernel_ms(), "-> us per chunk", prog.last_kernel_ms() * 1e3 / (n / 256))
