#!/usr/bin/env python3
"""bench.py — headline benchmark of the Dusp render path on MI355X.

Default workload (BASELINE.json configs[2], per-voice mode, SURVEY.md §8d "cfg3a"): 1024 independent voices per
GPU, voice k = Multiply(Osc(10*k), Ramp(T, 1, 0) triggered), T = 60 s at 48 kHz = 2 880 000 samples, every voice's
PCM written to HBM (4 B/sample, 11.8 GB per step per GPU).  A "step" is one render of the whole batch through the
C ABI (dusp_render_device): parameters and output stay resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config voices60|cfg4|cfg5] [--mode pcm|mixdown] [--gather]

  --config voices60  (default) weak scaling: --voices voices per GPU x --seconds
  --config cfg4      BASELINE configs[3]: 8192 feedback loops Osc -> Sum -> Delay(480) -> Filter(2000) -> Multiply(0.5) -> Sum x 10 s, sharded over the N GPUs
  --config cfg5      BASELINE configs[4]: 65 536 voices Multiply(Osc(20 + k/8), Ramp) x 1 s, sharded over the N GPUs
                     (strong scaling: the total is fixed)
  --mode mixdown     the voices of a rank go through the reference's left-deep Sum.many chain (Sum.js:18-29) into one
                     channel; ranks' partial mixes are reduced onto rank 0 (voice-samples/s; not HBM-bound by construction)
  --gather           also time render + RCCL gather of the PCM onto rank 0, tile-pipelined (reported, never `value`)

N > 1: one rank per GPU over RCCL.  Under torch.distributed.run the ranks are already there (RANK / WORLD_SIZE in the
environment); a bare `python bench.py --gpus N` starts them itself as child processes BEFORE this process touches the
GPU, relays rank 0's JSON line and exits with the children's status.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
XGMI_LINK_GBS = 153.0  # per-link peak, one link per peer (same guide)
# The JS reference itself, measured where it can run (the build container; it cannot travel to the GPU box): BASELINE.md §2
JS_REFERENCE = {"value": 7.0, "unit": "Msamples/s", "cores": 1, "kind": "reference",
                "hardware": "build container: Intel Xeon @ 2.10 GHz, 8 vCPU, 1 thread, Node v12.22.9",
                "sample": "1024 x Osc(10k) through the reference's renderChannelData, 1 s (BASELINE.md §2: 7.0 M voice-samples/s; "
                          "3.3 M for Multiply(Osc(Ramp), Osc) voices) — not measured on this box"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="voices60", choices=["voices60", "cfg4", "cfg5"])
    ap.add_argument("--mode", default="pcm", choices=["pcm", "mixdown"])
    ap.add_argument("--mix", default="reduce", choices=["chain", "reduce"],
                    help="mixdown on N > 1 GPUs: 'reduce' (default) = every rank's partial mix added onto rank 0 by an f32 reduction (tolerance-level); "
                         "'chain' = ONE left-deep chain continued rank to rank, window by window (bit for bit the single chain: shard.chain_mixdown; "
                         "covered by gloo ranks and by a replay of its posting order under RCCL's stream rules, but never run on two GPUs — opt-in until it has)")
    ap.add_argument("--seconds", type=float, default=None, help="rendered duration per voice (default: 60 / cfg5: 1)")
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU (voices60, default 1024) / in total (cfg5, default 65536)")
    ap.add_argument("--sample-rate", type=int, default=48000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-oracle baseline budget in seconds per leg (0 = skip)")
    ap.add_argument("--gather", action="store_true", help="also time render + gather of the PCM onto rank 0 (reported separately)")
    ap.add_argument("--gather-tile", type=int, default=0, help="voices per gather tile (0: shard / 8)")
    ap.add_argument("--engine", default="auto", choices=["auto", "chunk", "fused", "wave"])
    return ap.parse_args(argv)


def visible_gpus():
    """GPUs this process would see, counted WITHOUT loading the HIP runtime (no torch.cuda call, no HIP call): the KFD topology
    lists every compute node (`simd_count` > 0 = a GPU, 0 = a CPU), narrowed by the *_VISIBLE_DEVICES lists a launcher may have set."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for name in os.listdir(nodes):
            try:
                with open(os.path.join(nodes, name, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue  # (a node this user may not read is not one it can use)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children of a process that has not
    loaded the HIP runtime at all (devices are counted from the KFD topology in sysfs), wait, pass rank 0's line through."""
    import socket
    have = visible_gpus()
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this box" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def voice_program(cfg, sample_rate, n_samples, first_voice, n_voices):
    """Unified descriptor + parameter table for global voices [first_voice, first_voice + n_voices)."""
    import numpy as np
    import dusp_amd as d
    from dusp_amd import descriptor
    d.configure(sample_rate)
    if cfg == "cfg4":  # configs[3]: f_osc = 110 + k / 64, delay 480 of 4096, cutoff 2000, gain 0.5
        def loop(k):
            s = d.Sum(d.Osc(110.0 + k / 64.0), 0)
            f = d.Filter(d.Delay(s, 480, 4096), 2000)
            s.B = d.Multiply(f, 0.5)
            return f
        uni = descriptor.unify([descriptor.extract(loop(k)) for k in (0, 64)])
        assert uni.n_params == 1
        k = np.arange(first_voice, first_voice + n_voices, dtype=np.float64)
        return uni.words, (110.0 + k / 64.0).astype(np.float32).reshape(1, n_voices)
    freq = (lambda k: 10.0 * (k + 1)) if cfg == "voices60" else (lambda k: 20.0 + k / 8.0)
    # two representative circuits fix the structure; the parameter column is then written directly
    # (building 1024+ Python graphs just to read back f would only time the host)
    ex = [descriptor.extract(d.Multiply(d.Osc(freq(k)), d.Ramp(n_samples, 1, 0).trigger())) for k in (0, 1)]
    uni = descriptor.unify(ex)
    assert uni.n_params == 1
    k = np.arange(first_voice, first_voice + n_voices, dtype=np.float64)
    params = (10.0 * (k + 1) if cfg == "voices60" else 20.0 + k / 8.0).astype(np.float32)
    return uni.words, params.reshape(1, n_voices)


def mixdown_program(cfg, sample_rate, first_voice, n_voices):
    """One circuit: Sum.many of this rank's oscillators in voice order (constant f: the chain is one program, no parameters)."""
    import dusp_amd as d
    from dusp_amd import descriptor
    d.configure(sample_rate)
    freq = (lambda k: 10.0 * (k + 1)) if cfg == "voices60" else (lambda k: 20.0 + k / 8.0)
    mix = d.Sum.many([d.Osc(freq(k)) for k in range(first_voice, first_voice + n_voices)])
    return descriptor.extract(mix).words, None


def cpu_baseline(words, params, n_samples, budget_s):
    """Oracle (scalar C restatement of the reference, oracle/dusp_oracle.c) on a bounded sample of the same workload: whole
    voices, spread over the sweep — one thread until ~budget_s seconds of CPU are spent, then the same again with the
    voices spread over all the host cores this process may use (instances are independent; ctypes drops the GIL)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    oracle.build()
    n_inst = params.shape[1]
    order = np.random.RandomState(0).permutation(n_inst)

    def one(i):
        oracle.render(words, n_samples, params=params, n_instances=n_inst, instance=int(order[i % n_inst]), max_channels=1)

    done, t0 = 0, time.perf_counter()
    while done < n_inst and (done < 4 or time.perf_counter() - t0 < budget_s):
        one(done)
        done += 1
    dt = time.perf_counter() - t0
    host_cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host_cores
    res = {"value": round(done * n_samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%d of the %d voices x %d samples (%.1f s of CPU), one thread of %d host cores"
                     % (done, n_inst, n_samples, dt, host_cores)}
    threads = max(1, min(usable, 64))
    if threads > 1:
        per_voice = dt / done
        n_all = max(threads, min(n_inst, int(budget_s / per_voice) * threads))
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(one, range(n_all)))
        dta = time.perf_counter() - t1
        res["all_cores"] = {"value": round(n_all * n_samples / dta / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
                            "sample": "%d voices x %d samples over %d threads (%.1f s wall; %d cores visible, %d usable)"
                                      % (n_all, n_samples, threads, dta, host_cores, usable)}
    res["js_reference"] = JS_REFERENCE
    return res


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        self_launch(args)  # does not return

    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL prints its version banner to stdout when a
    # process group starts), so file descriptor 1 points at stderr from here on and the line goes out through a saved copy of the real one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from dusp_amd import runtime
    from dusp_amd.shard import chain_mixdown, instance_range, reduce_mixdown, render_and_gather

    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible); the render path has no CPU fallback" % (rank, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gather:
        # one rank: the same gather code runs (the root's own shard is a device copy, no peer), so that its timing and its
        # report are exercised on a one-GPU box too
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))

    sr = args.sample_rate
    cfg4 = args.config == "cfg4"
    cfg5 = args.config == "cfg5" or cfg4  # (both: a fixed set of instances, sharded)
    if cfg4 and args.mode == "mixdown":
        raise SystemExit("bench: --config cfg4 has no mix-down mode (every loop's PCM is written)")
    n_samples = int((args.seconds if args.seconds is not None else (10.0 if cfg4 else 1.0 if cfg5 else 60.0)) * sr)
    if cfg5:  # strong scaling: a fixed sweep, sharded
        n_total = args.voices or (8192 if cfg4 else 65536)
        lo, hi = instance_range(n_total, rank, world)
    else:     # weak scaling: V voices per GPU
        per = args.voices or 1024
        n_total = per * world
        lo, hi = rank * per, (rank + 1) * per
    n_voices = hi - lo
    mixdown = args.mode == "mixdown"

    os.environ.setdefault("DUSP_WAVE_JIT", "2")  # (a benchmark waits for a circuit's compiled kernel instead of rendering its first calls on the interpreter)
    ctx = runtime.Context(local_rank, sr)
    engine = {"auto": runtime.ENGINE_AUTO, "chunk": runtime.ENGINE_CHUNK, "fused": runtime.ENGINE_FUSED, "wave": runtime.ENGINE_WAVE}[args.engine]
    if mixdown:
        words, params = mixdown_program(args.config, sr, lo, n_voices)
        n_inst = 1
    else:
        words, params = voice_program(args.config, sr, n_samples, lo, n_voices)
        n_inst = n_voices
    prog = ctx.build(words, engine)
    d_params = torch.from_numpy(params).cuda() if params is not None else None
    d_out = torch.empty((n_inst, prog.n_out_channels, n_samples), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    p_ptr = d_params.data_ptr() if d_params is not None else None

    chained = mixdown and world > 1 and args.mix == "chain"
    # (chain: windows of the timeline travel down the ranks as a pipeline; ~4 windows per rank keep every GPU busy most of the time)
    chain_window = max(65536, (n_samples // (4 * world) + 2047) // 2048 * 2048)

    def chain_step():
        def render_window(first, n, init, raw, out):
            prog.render_chain_window(first, n, init.data_ptr() if init is not None else None, raw, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return chain_mixdown(render_window, n_samples, chain_window, d_out)

    def step():
        if chained:
            return chain_step()
        prog.render_device(n_samples, n_inst, p_ptr, d_out.data_ptr(), stream)
        if mixdown and world > 1:
            reduce_mixdown(d_out[0])

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
    mix_root = None
    for a, b in ev:
        a.record()
        if chained:
            mix_root = chain_step()  # (the events bracket the rank's whole step: renders of its windows and the waits between them)
            b.record()
            continue
        prog.render_device(n_samples, n_inst, p_ptr, d_out.data_ptr(), stream)
        b.record()
        if mixdown and world > 1:
            reduce_mixdown(d_out[0])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    # sanity: the timed output is real PCM (parity is the tests' job; here just refuse an all-zero / NaN buffer)
    head = (mix_root[0, :4096] if (chained and rank == 0) else d_out[0, 0, :4096] if not chained else torch.ones(8)).float().cpu().numpy()
    if not np.isfinite(head).all() or float(np.abs(head).max()) == 0.0:
        raise SystemExit("bench: rendered buffer is empty or non-finite")

    gather_info = None
    if args.gather and not mixdown:
        # the north star's "trivial RCCL gather of rendered PCM over xGMI": render + gather, tile-pipelined (tile k+1 renders
        # while round k is on the wire), timed on its own and never part of `value`
        tile = args.gather_tile or max(1, n_voices // 8)
        full = torch.empty((n_total, prog.n_out_channels, n_samples), dtype=torch.float32, device="cuda") if rank == 0 else None
        tiles = {}

        def render_tile(a, b):
            if (a, b) not in tiles:
                tiles[(a, b)] = d_params[:, a:b].contiguous()
            prog.render_device(n_samples, b - a, tiles[(a, b)].data_ptr(), d_out[a:b].data_ptr(), stream)

        times = []
        for it in range(3):
            barrier()
            g0 = time.perf_counter()
            render_and_gather(render_tile, d_out, n_total, tile=tile, full=full)
            barrier()
            times.append(time.perf_counter() - g0)
        gdt = min(times[1:])
        t = torch.tensor([gdt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gdt = float(t.item())
        if rank == 0:
            inbound = 4.0 * (n_total - n_voices) * prog.n_out_channels * n_samples
            ok = bool(torch.equal(full[lo:hi], d_out)) and bool(torch.isfinite(full[-1, 0, :4096]).all()) and float(full[-1].abs().max()) > 0
            gather_info = {"render_plus_gather_ms": round(gdt * 1e3, 3), "render_only_ms": round(float(np.mean(kernel_ms)), 4),
                           "inbound_GBps": round(inbound / gdt / 1e9, 1) if world > 1 else None,
                           "per_link_GBps": round(inbound / gdt / 1e9 / (world - 1), 1) if world > 1 else None,
                           "link_peak_GBps": XGMI_LINK_GBS, "tile_voices": tile, "rounds": (n_voices + tile - 1) // tile,
                           "Msamples_per_s_with_gather": round(float(n_total) * n_samples / gdt / 1e6, 1), "checked": ok,
                           "note": ("%d peers -> rank 0, one receive per peer and round posted together; round k overlaps the render of tile k+1" % (world - 1))
                                   if world > 1 else "one rank: no peer, the root's own shard is a device copy per tile (the gather's code path, not a link measurement)"}
        del full

    # The write ceiling of THIS box, for scale: a fill kernel (16-byte coalesced stores and nothing else) over the same
    # buffer, after the timed region.  The roofline fraction is against the 8 TB/s paper peak; HBM3E sustains less
    # for a pure write stream, and how much varies from box to box.
    fill_ms = []
    if rank == 0 and not mixdown:
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ctx.fill(d_out.data_ptr(), d_out.numel(), 0.25, stream)
            b.record()
            torch.cuda.synchronize()
            fill_ms.append(a.elapsed_time(b))

    if rank == 0:
        total_samples = float(n_total) * n_samples  # per step, all ranks (mix-down: voice-samples)
        ms_per_step = elapsed / args.steps * 1e3
        launch_ms = float(np.mean(kernel_ms))
        # per launch on this GPU: 4 B written per rendered sample, 0 read (the mix-down writes one channel)
        algo_bytes = 4.0 * n_inst * prog.n_out_channels * n_samples
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if not (name.startswith("traffic_") and name.endswith(".json")):
                continue
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            if rec.get("n_voices") == n_inst and rec.get("n_samples") == n_samples and rec.get("engine") == prog.engine and not mixdown:
                traffic = rec.get("hbm_bytes_per_launch") or rec.get("write_bytes_per_launch")  # (reads + writes where the kernel reads: configs[3]'s delay rings)
                traffic_source = "profiles/%s: %s" % (name, rec.get("source", "rocprofv3 --pmc WRITE_SIZE, own pass on an earlier box; not measured in this run"))
                break
        if mixdown:  # (mixdown_program: bare constant-f oscillators through Sum.many — no envelope)
            what = ("%d voices/GPU x Osc(10k)" % n_voices if not cfg5 else "%d voices in total x Osc(20+k/8)" % n_total)
        else:
            what = ("%d voices/GPU x Multiply(Osc(10k), Ramp(T,1,0) triggered)" % n_voices if not cfg5 else
                    "%d loops in total x Filter(Delay(Osc(110+k/64) + 0.5 x [the Filter's output], 480 of 4096), 2000)" % n_total if cfg4 else
                    "%d voices in total x Multiply(Osc(20+k/8), Ramp(T,1,0) triggered)" % n_total)
        line = {
            "metric": "rendered Msamples/sec (whole node) + HBM GB/s fraction, 1024-voice 48kHz",
            "value": round(total_samples * args.steps / elapsed / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if cfg5 else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %s: %s, %gs @%d Hz, %s"
                                   % ("configs[3]" if cfg4 else "configs[4]" if cfg5 else "configs[2]", "mix-down" if mixdown else "per-voice", what, n_samples / sr, sr,
                                      "voices folded through Sum.many into one channel per rank, partials reduced onto rank 0" if mixdown
                                      else "every voice's PCM written"),
                       "voices_per_gpu": n_voices, "voices_total": n_total, "n_samples": n_samples, "engine": prog.engine, "shape": prog.read_shape(),
                       "parallelism": "voices sharded over %d GPU(s), %s" % (world, ("one chain continued rank to rank, window by window (bit for bit)" if chained else "reduce of the partial mixes") if mixdown and world > 1 else "no collective")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": round(launch_ms, 4), "algorithmic_bytes_per_launch": algo_bytes},
        }
        if cfg4:
            line["roofline"]["note"] = ("the Filter's recurrence bounds this circuit, not HBM: as a scan over the chunk (default; the gate's bound for this circuit is "
                                        "2^-24 (sum|h| + 2) / (1 - loop gain 0.5) = 2.44e-6 of the Filter's output scale, measured 3.5e-7: DESIGN.md 6.2c) the kernel is "
                                        "bound by vector-ALU issue (201 instructions a wavefront and chunk, 91 % of the kernel's cycles: profiles/r04_cfg4_pmc.md); the Delay's 480 samples "
                                        "are a line of its input in LDS (no ring in memory: HBM traffic = the PCM written); DUSP_FILTER_SCAN=0 renders the oracle's bits on the Filter stage")
        if mixdown:
            line["roofline"]["note"] = "mix-down writes one channel: ALU/LDS-bound by construction, not graded against HBM (SURVEY.md §8d)"
        if fill_ms:
            ceiling = algo_bytes / (min(fill_ms[1:]) * 1e-3) / 1e9
            line["roofline"]["write_ceiling_measured"] = {"GBps": round(ceiling, 1), "frac_of_it": round(achieved / ceiling, 4),
                                                          "what": "dusp_fill_device over the same buffer on this box, best of 3"}
        if world == 1 and args.cpu_seconds > 0 and not mixdown:
            line["cpu_baseline"] = cpu_baseline(words, params, n_samples, args.cpu_seconds)
        # (fixed schema: the gather's fields are always there; null where --gather was not asked for)
        line["gather"] = gather_info or {"render_plus_gather_ms": None, "render_only_ms": None, "inbound_GBps": None, "per_link_GBps": None,
                                          "link_peak_GBps": XGMI_LINK_GBS, "tile_voices": None, "rounds": None,
                                          "Msamples_per_s_with_gather": None, "checked": None, "note": "not measured: run with --gather"}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
