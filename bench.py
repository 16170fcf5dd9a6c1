#!/usr/bin/env python3
"""bench.py — headline benchmark of the Dusp render path on MI355X.

Workload (BASELINE.json configs[2], per-voice mode, SURVEY.md §8d "cfg3a"): 1024
independent voices per GPU, voice k = Multiply(Osc(10*k), Ramp(T, 1, 0) triggered),
T = 60 s at 48 kHz = 2 880 000 samples, every voice's PCM written to HBM
(4 B/sample, 11.8 GB per step per GPU).  A "step" is one render of the whole batch
through the C ABI (dusp_render_device): parameters and output stay resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--seconds S] [--voices V]

N > 1 is launched by torch.distributed.run (one rank per GPU); voices shard across ranks with no
data-path collective (weak scaling: V voices per GPU).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")


def voice_program(sample_rate, n_samples, n_voices, first_voice):
    """Unified descriptor + parameter table for voices first_voice+1 .. first_voice+n_voices."""
    import dusp_amd as d
    from dusp_amd import descriptor
    d.configure(sample_rate)
    # two representative circuits fix the structure; the parameter column is then written directly
    # (building 1024+ Python graphs just to read back f = 10k would only time the host)
    ex = [descriptor.extract(d.Multiply(d.Osc(10 * k), d.Ramp(n_samples, 1, 0).trigger())) for k in (1, 2)]
    uni = descriptor.unify(ex)
    assert uni.n_params == 1
    params = (10.0 * np.arange(first_voice + 1, first_voice + n_voices + 1, dtype=np.float64)).astype(np.float32)
    return uni.words, params.reshape(1, n_voices)


def cpu_baseline(words, params, n_samples, budget_s):
    """Oracle (scalar C restatement of the reference, oracle/dusp_oracle.c) on a bounded sample of the
    same workload: whole voices, spread over the sweep, until ~budget_s seconds of CPU have been spent."""
    from oracle import oracle
    oracle.build()
    n_inst = params.shape[1]
    order = np.random.RandomState(0).permutation(n_inst)
    done, t0 = 0, time.perf_counter()
    while done < n_inst and (done < 4 or time.perf_counter() - t0 < budget_s):
        oracle.render(words, n_samples, params=params, n_instances=n_inst, instance=int(order[done]), max_channels=1)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(done * n_samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d voices x %d samples (%.1f s of CPU), one thread of %d host cores"
                      % (done, n_inst, n_samples, dt, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=60.0, help="rendered duration per voice")
    ap.add_argument("--voices", type=int, default=1024, help="voices per GPU")
    ap.add_argument("--sample-rate", type=int, default=48000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-oracle baseline budget in seconds (0 = skip)")
    ap.add_argument("--gather", action="store_true", help="also time an RCCL gather of the PCM onto rank 0 (reported separately)")
    ap.add_argument("--engine", default="auto", choices=["auto", "chunk", "fused"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dusp_amd import runtime

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch N>1 with torch.distributed.run" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sr = args.sample_rate
    n_samples = int(args.seconds * sr)
    n_voices = args.voices
    words, params = voice_program(sr, n_samples, n_voices, first_voice=rank * n_voices)

    ctx = runtime.Context(local_rank, sr)
    engine = {"auto": runtime.ENGINE_AUTO, "chunk": runtime.ENGINE_CHUNK, "fused": runtime.ENGINE_FUSED}[args.engine]
    prog = ctx.build(words, engine)
    d_params = torch.from_numpy(params).cuda()
    d_out = torch.empty((n_voices, prog.n_out_channels, n_samples), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        prog.render_device(n_samples, n_voices, d_params.data_ptr(), d_out.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # kernel time: HIP events on the launch stream, one pair per step
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    # sanity: the timed output is real PCM (voice 0 of this rank, first samples, against the closed form is
    # the tests' job; here just refuse an all-zero / NaN buffer)
    head = d_out[0, 0, :4096].float().cpu().numpy()
    if not np.isfinite(head).all() or float(np.abs(head).max()) == 0.0:
        raise SystemExit("bench: rendered buffer is empty or non-finite")

    gather_info = None
    if args.gather and world > 1:
        # the north star's "trivial RCCL gather of rendered PCM over xGMI": timed on its own, never part of `value`
        from dusp_amd.shard import gather_pcm
        seen = [0]

        def sink(lo, hi, t):
            seen[0] += int(t.numel())

        barrier()
        g0 = time.perf_counter()
        # every rank holds global instances [rank*V, (rank+1)*V): gather them all, tile by tile, onto rank 0
        gather_pcm(d_out, n_voices * world, tile=16, sink=sink)
        barrier()
        gdt = time.perf_counter() - g0
        if rank == 0:
            gbytes = 4.0 * n_voices * n_samples * (world - 1)
            gather_info = {"seconds": round(gdt, 4), "inbound_GBps": round(gbytes / gdt / 1e9, 1),
                           "note": "PCM of %d peers sent point-to-point to rank 0 in 16-voice tiles" % (world - 1)}

    # The write ceiling of THIS box, for scale: a fill kernel (16-byte coalesced stores and nothing else) over the same
    # buffer, after the timed region.  The roofline fraction above is against the 8 TB/s paper peak; HBM3E sustains less
    # for a pure write stream, and how much varies from box to box.
    fill_ms = []
    if rank == 0:
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ctx.fill(d_out.data_ptr(), d_out.numel(), 0.25, stream)
            b.record()
            torch.cuda.synchronize()
            fill_ms.append(a.elapsed_time(b))

    if rank == 0:
        total_samples = float(n_voices) * n_samples * world
        ms_per_step = elapsed / args.steps * 1e3
        launch_ms = float(np.mean(kernel_ms))
        algo_bytes = 4.0 * n_voices * n_samples  # per launch on this GPU: 4 B written per rendered sample, 0 read
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tf):
            with open(tf) as f:
                rec = json.load(f)
            if rec.get("n_voices") == n_voices and rec.get("n_samples") == n_samples and rec.get("engine") == prog.engine:
                traffic = rec.get("write_bytes_per_launch")
        line = {
            "metric": "rendered Msamples/sec (whole node) + HBM GB/s fraction, 1024-voice 48kHz",
            "value": round(total_samples * args.steps / elapsed / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2] per-voice: %d voices/GPU x Multiply(Osc(10k), Ramp(T,1,0) triggered), "
                                   "%gs @%d Hz, every voice's PCM written" % (n_voices, args.seconds, sr),
                       "voices_per_gpu": n_voices, "n_samples": n_samples, "engine": prog.engine, "shape": prog.shape,
                       "parallelism": "voices sharded over %d GPU(s), no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(launch_ms, 4), "algorithmic_bytes_per_launch": algo_bytes},
        }
        if fill_ms:
            ceiling = algo_bytes / (min(fill_ms[1:]) * 1e-3) / 1e9
            line["roofline"]["write_ceiling_measured"] = {"GBps": round(ceiling, 1), "frac_of_it": round(achieved / ceiling, 4),
                                                          "what": "dusp_fill_device over the same buffer on this box, best of 3"}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(words, params, n_samples, args.cpu_seconds)
        if gather_info:
            line["gather"] = gather_info
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
