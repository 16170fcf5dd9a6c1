#!/usr/bin/env python3
"""A/B micro-benchmarks in ONE process (cdna guide §5.4 rule 24): fill-kernel ceiling vs the fused
render kernel under its variants.  Prints one line per variant: median / min kernel ms and GB/s."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--globaltbl", action="store_true")
    args = ap.parse_args()
    import torch
    import dusp_amd as d
    from dusp_amd import descriptor, runtime

    sr = 48000
    d.configure(sr)
    n = int(args.seconds * sr)
    V = args.voices
    ctx = runtime.Context(0, sr)
    out = torch.empty((V, 1, n), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    nbytes = 4.0 * V * n

    def graph(kind, f):
        if kind == "osc":
            return d.Osc(f)
        if kind == "oscramp":
            return d.Multiply(d.Osc(f), d.Ramp(n, 1, 0).trigger())
        if kind == "oscgain":
            return d.Multiply(d.Osc(f), 0.5)

    variants = []
    envsets = [("nt", {}), ("nt/plainfx", {"DUSP_FUSED_FX32": "2"}), ("nt/r8", {"DUSP_FUSED_R": "8"}), ("nt/it2", {"DUSP_FUSED_ITEMS": "2"})]
    KNOBS = ["DUSP_FUSED_R", "DUSP_FUSED_FX32", "DUSP_FUSED_ITEMS", "DUSP_FUSED_TABLE", "DUSP_FUSED_SEGMAJOR"]

    def context_under(env):  # the library reads its A/B knobs once, when a context is created
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        return runtime.Context(0, sr)

    ctxs = [(ename, context_under(env)) for ename, env in envsets]
    for kind in ["osc", "oscramp"]:
        for fname, fs in [("int", 10.0 * np.arange(1, V + 1)), ("frac", 20 + np.arange(V) / 8.0)]:
            uni = descriptor.unify([descriptor.extract(graph(kind, float(f))) for f in fs[:2]])
            params = torch.from_numpy(fs.astype(np.float32).reshape(1, V)).cuda()
            for ename, c in ctxs:
                variants.append(("%s/%s/%s" % (kind, fname, ename), c.build(uni.words), params, None))

    def run(v):
        name, prog, params, env = v
        if prog is None:
            ctx.fill(out.data_ptr(), out.numel(), 1.0, stream)
        else:
            prog.render_device(n, V, params.data_ptr(), out.data_ptr(), stream)

    variants.insert(0, ("fill", None, None, None))
    times = {v[0]: [] for v in variants}
    for r in range(args.rounds + 1):
        for v in variants:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            run(v)
            b.record()
            torch.cuda.synchronize()
            if r > 0:
                times[v[0]].append(a.elapsed_time(b))
    for name, ts in times.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        print("%-30s median %8.3f ms  min %8.3f ms   %7.1f GB/s (median)  %5.1f%% of 8 TB/s" %
              (name, med, mn, nbytes / med / 1e6, nbytes / med / 1e6 / 80.0), flush=True)


if __name__ == "__main__":
    main()
