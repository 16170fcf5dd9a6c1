#!/usr/bin/env python3
"""Time every BASELINE.json config on one GPU (kernel time by HIP events; PCM stays in HBM).
  configs[1] cfg2   Multiply(Osc(f: Ramp), Osc(3)), 10 s, 1 instance (literal + sweep readings)
  configs[2] cfg3a  1024 voices Multiply(Osc(10k), Ramp), 60 s, per-voice PCM   (= bench.py)
             cfg3b  the same 1024 Osc(10k) through Sum.many, mix-down to one channel
  configs[3] cfg4   8192 instances of the Osc->Sum->Delay->Filter->Multiply feedback loop, 10 s
  configs[4] cfg5   65536-voice sweep Multiply(Osc(20+k/8), Ramp): one GPU's shard = 8192 voices x 1 s
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DUSP_WAVE_JIT", "2")  # wait for every circuit's compiled kernel


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="scale durations (for quick runs)")
    ap.add_argument("--json", default="", help="write one record per config (for tools/profile_summary.py)")
    args = ap.parse_args()
    import torch
    import dusp_amd as d
    from dusp_amd import descriptor, runtime
    sr = 48000
    d.configure(sr)
    ctx = runtime.Context(0, sr)
    stream = torch.cuda.current_stream().cuda_stream

    def loop(k, delay=480, cutoff=2000):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, delay, 4096), cutoff)
        s.B = d.Multiply(f, 0.5)
        return f

    def uni_of(graphs):
        return descriptor.unify([descriptor.extract(g) for g in graphs])

    FILL = box_fill_GBps(ctx, torch, stream)
    print("fill kernel on this box: %.0f GB/s" % FILL, flush=True)
    cfgs = {}
    records = []
    T10 = int(10 * sr * args.scale)
    T60 = int(60 * sr * args.scale)
    T1 = int(1 * sr * args.scale)
    cfgs["cfg2_literal"] = (lambda: uni_of([d.Multiply(d.Osc(d.Ramp(200, 100, 2)), d.Osc(3))]), T10, None)
    cfgs["cfg2_sweep"] = (lambda: uni_of([d.Multiply(d.Osc(d.Ramp(2 * sr, 200, 100).trigger()), d.Osc(3))]), T10, None)
    cfgs["cfg3a"] = (lambda: uni_of([d.Multiply(d.Osc(10 * k), d.Ramp(T60, 1, 0).trigger()) for k in (1, 2)]), T60,
                     (10.0 * np.arange(1, 1025)).astype(np.float32).reshape(1, -1))
    cfgs["cfg3b_sum1024"] = (lambda: uni_of([d.Sum.many([d.Osc(10 * k) for k in range(1, 1025)])]), T60, None)
    # ... the same 1024 voices under their envelopes (configs[2]'s voice as BASELINE spells it), folded by Sum.many: the fused sum chain's enveloped form
    cfgs["cfg3c_sum1024_enveloped"] = (lambda: uni_of([d.Sum.many([d.Multiply(d.Osc(10 * k), d.Ramp(T60, 1, 0).trigger()) for k in range(1, 1025)])]), T60, None)
    cfgs["cfg4_loop8192"] = (lambda: uni_of([loop(k) for k in (0, 64)]), T10,
                             (110 + np.arange(8192) / 64.0).astype(np.float32).reshape(1, -1))
    # a delay-time sweep of the configs[3] voice: delay = 300 + k % 400 samples per instance (parameter rows: f, delay — in unit order)
    cfgs["cfg4_delay_sweep"] = (lambda: uni_of([loop(k, 300 + k % 400) for k in (0, 65)]), T10, "delay_sweep")
    # a cutoff sweep of the same voice: cutoff = 2000 + k % 4000 Hz per instance (the whole column passes the scan's gate: one launch looks at it)
    cfgs["cfg4_cutoff_sweep"] = (lambda: uni_of([loop(k, 480, 2000 + k % 4000) for k in (0, 65)]), T10, "cutoff_sweep")
    cfgs["cfg5_shard8192"] = (lambda: uni_of([d.Multiply(d.Osc(20 + k / 8), d.Ramp(T1, 1, 0).trigger()) for k in (0, 1)]), T1,
                              (20 + np.arange(8192) / 8.0).astype(np.float32).reshape(1, -1))
    for name, cfg in cfgs.items():
        build, n, params = cfg[:3]
        engine = cfg[3] if len(cfg) > 3 else runtime.ENGINE_AUTO
        if args.only and name not in args.only.split(","):
            continue
        t0 = time.time()
        uni = build()
        if isinstance(params, str):  # delay / cutoff sweep: which row is f, which the swept value, is read off the two circuits
            k = np.arange(8192)
            rows = []
            swept = (300 + k % 400) if params == "delay_sweep" else (2000 + k % 4000)
            for p in range(uni.n_params):
                rows.append((110 + k / 64.0) if abs(float(uni.params[p, 0]) - 110.0) < 1e-3 else swept.astype(np.float64))
            params = np.ascontiguousarray(np.stack(rows).astype(np.float32))
        prog = ctx.build(uni.words, engine)
        n_inst = params.shape[1] if params is not None else 1
        dp = torch.from_numpy(params).cuda() if params is not None else None
        out = torch.empty((n_inst, prog.n_out_channels, n), dtype=torch.float32, device="cuda")
        host_s = time.time() - t0
        ts = []
        for r in range(args.rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            prog.render_device(n, n_inst, dp.data_ptr() if dp is not None else None, out.data_ptr(), stream)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        prog._read_info()
        samples = float(n_inst) * prog.n_out_channels * n
        print("%-16s engine=%-5s %-18s inst=%-6d n=%-8d  %10.3f ms  %12.1f Msamples/s out  %8.1f GB/s  (host build %.2fs)"
              % (name, prog.engine, prog.shape, n_inst, n, ms, samples / ms / 1e3, 4 * samples / ms / 1e6, host_s), flush=True)
        records.append({"config": name, "engine": prog.engine, "shape": prog.shape, "kernel": "dusp_jit_render" if "compiled kernel" in prog.shape else "dusp_%s_kernel" % prog.engine,
                        "instances": n_inst, "channels": prog.n_out_channels, "n_samples": n, "avg_ms": round(ms, 4),
                        "first_render_ms_compile_inclusive": round(ts[0], 3), "algorithmic_bytes": 4 * samples, "frac_of_8TBps": round(4 * samples / (ms * 1e-3) / 8e12, 4),
                        "frac_of_fill": round(4 * samples / (ms * 1e-3) / 1e9 / FILL, 4), "box_fill_GBps": round(FILL, 1)})
        prog.close()
        del out
    if args.json:
        json.dump(records, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
