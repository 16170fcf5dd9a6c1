#!/usr/bin/env python3
"""Per-unit cost on the wave engine: small graphs forced onto ENGINE_WAVE, 16384 instances x 1 s (time split off), and patch
topologies (descriptors extracted from the reference's own patch objects: tests/golden/patch_*) x 1024 instances.
  python tools/wave_ops.py ["--only=name;name"] [--json=path]     (--json: one record per graph, for tools/profile_summary.py)"""
import json
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
def box_fill_GBps(ctx, torch, stream):
    """What a pure write stream sustains on THIS box: dusp_fill_device (16-byte coalesced stores) over 4 GiB, best of 4."""
    buf = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.fill(buf.data_ptr(), buf.numel(), 0.25, stream)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    del buf
    return 4.0 * (1 << 30) / (best * 1e-3) / 1e9


ctx = runtime.Context(0, sr)
stream = torch.cuda.current_stream().cuda_stream
V, n = 16384, 48000
FILL = box_fill_GBps(ctx, torch, stream)
print("fill kernel on this box: %.0f GB/s" % FILL, flush=True)
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
    "filter(osc, 3 kHz)": lambda k: d.Filter(d.Osc(110 + k / 8), 3000),  # (above the scan's bound: JitFilterScan, no Filter stage — DESIGN.md §6.2c)
    "allpass(osc)": lambda k: (lambda a: (setattr(a, "IN", d.Osc(110 + k / 8)), a)[1])(d.AllPass(0.0021, 0.6)),
    "filter(osc) * ramp": lambda k: d.Multiply(d.Filter(d.Osc(110 + k / 8), 800 + k / 16), d.Ramp(sr, 1, 0).trigger()),
    "filter(osc, lfo)": lambda k: d.Filter(d.Osc(110 + k / 8), d.Sum(d.Multiply(d.Osc(5), 800), 1000)),
    "delay(osc, 300)": lambda k: d.Delay(d.Osc(110 + k / 8), 300, 2048),  # (a whole delay: the input kept as a line in LDS, no ring in memory — JitDelayLine)
    "delay(osc, 1001)": lambda k: d.Delay(d.Osc(110 + k / 8), 1001, 2048),
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
json_path = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--json=")] or [None])[0]
records = []
# patch topologies: the reference's own patches as extracted descriptors (constants only: every instance renders the same circuit)
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
PATCHES = ["patch_fm_osc", "patch_simple_delay", "patch_multitap", "patch_many_osc", "patch_stereo_detune", "patch_space",
           "rt_shape_fast", "rt_dev_rhythm"]  # (+ two circuits with device Retriggerers: an enveloped oscillator retriggered at 50 Hz; two envelopes into a feedback delay line)
for name in PATCHES:
    graphs["%s x 256" % name] = name
for name, g in graphs.items():
    if only and name not in only[0].split(";"):
        continue
    if isinstance(g, str):
        words = np.fromfile(os.path.join(GOLDEN, g + ".desc.f64"), dtype=np.float64)
        V_, dp, params = 256, None, None  # (some of these patches own seconds of delay line per instance: patch_multitap's CircleBuffer is 230 M samples —
        # 236 GB of rings for 256 instances; a render zero-fills the part of them it can reach (dusp_abi.hip zero_rings): 38 ms with the whole fill, 1.1 ms now)
        prog = ctx.build(words, runtime.ENGINE_AUTO)
    else:
        full = descriptor.unify([descriptor.extract(g(k)) for k in (0, 8, 16)])
        base = full.params[:, 0].astype(np.float64)
        step = (full.params[:, 1].astype(np.float64) - base) / 8.0
        params = (base[:, None] + step[:, None] * np.arange(V)[None, :]).astype(np.float32)
        dp = torch.from_numpy(np.ascontiguousarray(params)).cuda()
        V_ = V
        prog = ctx.build(full.words, runtime.ENGINE_WAVE)
    out = torch.empty((V_, prog.n_out_channels, n), dtype=torch.float32, device="cuda")
    ts = []
    try:
        for r in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            prog.render_device(n, V_, dp.data_ptr() if dp is not None else None, out.data_ptr(), stream)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
    except runtime.DuspHipError as e:
        print("%-28s failed: %s" % (name, e), flush=True)
        prog.close()
        del out
        continue
    ms = float(np.median(ts))
    first_ms = ts[0]
    prog._read_info()
    algo = 4.0 * V_ * prog.n_out_channels * n
    print("%-28s %8.3f ms  %8.1f Gsamples/s  %5.1f %% of 8 TB/s   [%s %s]" % (name, ms, V_ * prog.n_out_channels * n / ms / 1e6, 100 * algo / (ms * 1e-3) / 8e12, prog.engine, prog.shape), flush=True)
    records.append({"graph": name, "kernel": "dusp_jit_render" if "compiled kernel" in prog.shape else prog.engine, "engine": prog.engine, "shape": prog.shape,
                    "instances": V_, "channels": prog.n_out_channels, "n_samples": n, "avg_ms": round(ms, 4), "first_render_ms_compile_inclusive": round(first_ms, 3),
                    "algorithmic_bytes": algo, "frac_of_8TBps": round(algo / (ms * 1e-3) / 8e12, 4),
                    "frac_of_fill": round(algo / (ms * 1e-3) / 1e9 / FILL, 4), "box_fill_GBps": round(FILL, 1)})
    prog.close()
    del out
if json_path:
    json.dump(records, open(json_path, "w"), indent=1)
