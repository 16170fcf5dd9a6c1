# tests/measure_filter_scan_error.py — how far the scan form of a Filter (jit_prelude.hpp JitFilterScan) is from the oracle and from the Filter stage: ten seconds per case (the oracle as the checker; run on the GPU box: profiles/r03_filter_scan.txt)
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
import dusp_amd as d
from dusp_amd import descriptor, render, runtime
from conftest import knob_context
from oracle import oracle
oracle.build()
d.configure(48000)
def loop(k, cutoff=2000):
    s = d.Sum(d.Osc(110 + k / 64), 0)
    f = d.Filter(d.Delay(s, 480, 4096), cutoff)
    s.B = d.Multiply(f, 0.5)
    return f
for name, build in (("loop 2000", loop), ("lp1600 saw", lambda k: d.Filter(d.Osc(220 + k, "saw"), 1600)), ("hp5000 saw", lambda k: d.Filter(d.Osc(220 + k, "saw"), 5000, "HP"))):
    uni = descriptor.unify([descriptor.extract(build(k)) for k in (0, 64)])
    n = 480000
    out = {}
    for scan in (1, 0):
        prog = knob_context(48000, DUSP_FILTER_SCAN=scan).build(uni.words, runtime.ENGINE_WAVE)
        out[scan] = prog.render(n, 2, uni.params)[1, 0].astype(np.float64)
        prog.close()
    want = oracle.render(uni.words, n, params=uni.params, n_instances=2, instance=1)[0].astype(np.float64)
    sc = np.max(np.abs(want))
    print("%-12s scale %.3f  scan vs oracle: max %.2e rms %.2e (of scale)   stage vs oracle: max %.2e   samples that differ from the oracle: scan %.1f %%, stage %.1f %%" % (
        name, sc, np.max(np.abs(out[1] - want)) / sc, np.sqrt(np.mean((out[1] - want) ** 2)) / sc, np.max(np.abs(out[0] - want)) / sc,
        100 * np.mean(out[1] != want), 100 * np.mean(out[0] != want)))
