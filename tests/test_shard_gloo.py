"""N>1 path on CPU: gloo ranks shard the voices, each 'renders' its own range tile by tile, rank 0 gathers.

The GPU box runs the same host logic with backend 'nccl' (RCCL); here the renderer is the CPU oracle
(test infrastructure) so that the gathered PCM can be checked against a single-process render.  The HIP
renderer goes through the very same code at world_size 1 in test_sharded_render_on_the_gpu (-m gpu)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from dusp_amd.shard import instance_range, n_tile_rounds


def test_instance_range_partitions_exactly():
    for n in (1, 7, 64, 1000, 65536):
        for world in (1, 2, 3, 4, 8):
            spans = [instance_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            assert n_tile_rounds(n, world, 5) == (max(sizes) + 4) // 5


def _voices(n_voices, n_samples):
    import dusp_amd as d
    from dusp_amd import descriptor
    d.configure(48000)
    return descriptor.unify([descriptor.extract(d.Multiply(d.Osc(20 + k / 8), d.Ramp(n_samples, 1, 0).trigger()))
                             for k in range(n_voices)])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_voices, n_samples, tile, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dusp_amd.shard import gather_pcm, instance_range, reduce_mixdown, render_and_gather
    from oracle import oracle
    uni = _voices(n_voices, n_samples)
    lo, hi = instance_range(n_voices, rank, world)
    local = torch.zeros((hi - lo, 1, n_samples), dtype=torch.float32)
    rendered = []

    def render_tile(a, b):  # what the GPU box does with dusp_render_device on voices [lo+a, lo+b)
        rendered.append((a, b))
        for i in range(a, b):
            local[i] = torch.from_numpy(oracle.render(uni.words, n_samples, params=uni.params, n_instances=n_voices, instance=lo + i))

    full = render_and_gather(render_tile, local, n_voices, tile=tile)
    assert rendered == [(a, min(a + tile, hi - lo)) for a in range(0, hi - lo, tile)]
    again = gather_pcm(local, n_voices, tile=tile + 1)  # one-shot gather of an already rendered shard, other tiling
    seen = []
    gather_pcm(local, n_voices, tile=tile, sink=lambda a, b, t: seen.append((a, b, t.clone())))
    mix = reduce_mixdown(local.sum(dim=0))  # [1, n_samples] partial -> sum over ranks on the root
    if rank == 0:
        assert torch.equal(full, again)
        via_sink = torch.zeros_like(full)
        for a, b, t in seen:
            via_sink[a:b] = t
        assert torch.equal(full, via_sink) and sum(b - a for a, b, _ in seen) == n_voices
        np.save(result_path, full.numpy())
        np.save(result_path + ".mix.npy", mix.numpy())
    else:
        assert full is None and again is None and mix is None and not seen
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_voices,tile", [(2, 7, 2), (3, 8, 1)])
def test_ranks_shard_render_and_gather(tmp_path, oracle, world, n_voices, tile):
    n_samples = 700
    result = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_voices, n_samples, tile, result), nprocs=world, join=True)
    got = np.load(result)
    uni = _voices(n_voices, n_samples)
    want = np.stack([oracle.render(uni.words, n_samples, params=uni.params, n_instances=n_voices, instance=i)
                     for i in range(n_voices)])
    assert got.shape == want.shape and np.array_equal(got, want)
    mix = np.load(result + ".mix.npy")
    assert np.allclose(mix, want.astype(np.float64).sum(axis=0), rtol=0, atol=1e-5 * n_voices)


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` starts its own ranks — but only after counting devices, without touching one."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_sharded_render_on_the_gpu(oracle):
    """world_size 1 on the GPU box: the HIP renderer driven by the sharding code (tile-pipelined render + gather), bit-exact
    against the oracle; then `bench.py --gpus 2` on this one-GPU box must fail cleanly."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    from dusp_amd import runtime
    from dusp_amd.shard import render_and_gather
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n_voices, n_samples, tile = 37, 5000, 8
        uni = _voices(n_voices, n_samples)
        ctx = runtime.Context(0, 48000)
        prog = ctx.build(uni.words)
        d_params = torch.from_numpy(uni.params).cuda()
        local = torch.zeros((n_voices, 1, n_samples), dtype=torch.float32, device="cuda")
        keep = []

        def render_tile(a, b):
            p = d_params[:, a:b].contiguous()
            keep.append(p)
            prog.render_device(n_samples, b - a, p.data_ptr(), local[a:b].data_ptr(), torch.cuda.current_stream().cuda_stream)

        full = render_and_gather(render_tile, local, n_voices, tile=tile)
        torch.cuda.synchronize()
        got = full.cpu().numpy()
        for i in range(n_voices):
            want = oracle.render(uni.words, n_samples, params=uni.params, n_instances=n_voices, instance=i)
            assert np.array_equal(got[i], want), "voice %d" % i
    finally:
        dist.destroy_process_group()
    if torch.cuda.device_count() == 1:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300,
                           env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
        assert r.returncode != 0 and "only 1 GPU(s) visible" in (r.stderr + r.stdout)


def test_the_launcher_counts_gpus_without_the_hip_runtime():
    """bench.py's device count reads the KFD topology from sysfs: no torch import, no HIP call in the parent of the ranks."""
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.visible_gpus(); "
            "assert 'torch' not in sys.modules, 'the launcher imported torch'; print(n)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert int(r.stdout.strip()) >= 0
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0 and int(r.stdout.strip()) == 0, r.stderr


@pytest.mark.gpu
def test_bench_gather_runs_at_one_rank():
    """`bench.py --gpus 1 --config cfg5 --gather`: the gather's timing and report code on a one-GPU box (self-copy, no peer);
    and the sysfs device count agrees with what the HIP runtime sees."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "cfg5", "--voices", "4096", "--seconds", "0.25",
                        "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--gather"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["gather"]["checked"] is True and line["gather"]["render_plus_gather_ms"] > 0 and line["gather"]["rounds"] == 8
    assert line["n_gpus"] == 1 and line["config"]["voices_total"] == 4096
    sys.path.insert(0, ROOT)
    import bench
    assert bench.visible_gpus() == torch.cuda.device_count()


@pytest.mark.gpu
def test_bench_renders_the_feedback_loops_of_configs_3():
    """`bench.py --config cfg4` (BASELINE configs[3], a reduced set): the line names the workload, carries roofline and cpu_baseline."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "cfg4", "--voices", "512", "--seconds", "0.5",
                        "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["workload"].startswith("configs[3] per-voice: 512 loops") and line["config"]["engine"] == "wave"
    assert "compiled kernel" in line["config"]["shape"] and line["scaling"] == "strong" and line["value"] > 0
    assert line["roofline"]["algorithmic_bytes_per_launch"] == 4.0 * 512 * 24000 and line["cpu_baseline"]["value"] > 0


# ---- one Sum.many chain over several ranks, bit for bit (shard.chain_mixdown / dusp_render_chain_window)

def _chain_voice_freqs(n_voices):
    return [55.0 + 13.25 * k for k in range(n_voices)]


def _chain_worker(rank, world, port, n_voices, n_samples, window, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dusp_amd as d
    from dusp_amd import descriptor
    from dusp_amd.shard import chain_mixdown, instance_range
    from oracle import oracle
    d.configure(48000)
    lo, hi = instance_range(n_voices, rank, world)
    freqs = _chain_voice_freqs(n_voices)
    # this rank's voices, each rendered by itself (what the sum chain kernel looks up), WITHOUT the copy-out's `x || 0` mattering: sines
    mine = [oracle.render(descriptor.extract(d.Osc(f)).words, n_samples)[0] for f in freqs[lo:hi]]
    calls = []

    def render_window(first, n, init, raw, out):  # the CPU stand-in of Program.render_chain_window: left-deep f32 adds (Sum.js:18-29)
        calls.append((first, n, init is not None, raw))
        acc = init[0].numpy().copy() if init is not None else np.zeros(n, dtype=np.float32)
        for v in mine:
            acc = (acc + v[first:first + n]).astype(np.float32)
        if not raw:
            acc = np.where(np.isnan(acc), np.float32(0), acc) + np.float32(0)
        out[0] = torch.from_numpy(acc)

    # the order in which this rank posts its transfers and waits for them, for the replay under RCCL's rules (below)
    log, real_batch = [], dist.batch_isend_irecv

    class Logged:
        def __init__(self, work, ident):
            self.work, self.ident = work, ident

        def wait(self):
            log.append(("wait", self.ident))
            return self.work.wait()

    def logging_batch(ops):
        assert len(ops) == 1
        op = ops[0]
        ident = len(log)
        log.append(("post", ident, "send" if op.op is dist.isend else "recv", op.peer, "default" if (op.group is None or op.group is dist.group.WORLD) else "return"))
        return [Logged(w, ident) for w in real_batch(ops)]

    dist.batch_isend_irecv = logging_batch
    try:
        mix = chain_mixdown(render_window, n_samples, window, torch.zeros(1), group=None)
    finally:
        dist.batch_isend_irecv = real_batch
    spans = [(a, min(a + window, n_samples) - a) for a in range(0, n_samples, window)]
    assert calls == [(a, n, rank > 0, rank != world - 1) for a, n in spans]
    logs = [None] * world
    dist.all_gather_object(logs, log)
    if rank == 0:
        np.save(result_path, mix.numpy())
        with open(result_path + ".posts.json", "w") as f:
            json.dump(logs, f)
    else:
        assert mix is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_voices,n_samples,window", [(2, 7, 5000, 2048), (3, 8, 4096, 2048), (1, 3, 3000, 2048)])
def test_one_sum_chain_over_several_ranks_is_the_single_chain_bit_for_bit(tmp_path, oracle, world, n_voices, n_samples, window):
    """shard.chain_mixdown on gloo ranks: every rank continues, window by window, the running sums the rank before left; the root's
    mix equals the single-process render of the whole `Sum.many` (the oracle: the reference's left-deep f32 chain) bit for bit."""
    import dusp_amd as d
    from dusp_amd import descriptor
    result = str(tmp_path / "mix.npy")
    mp.spawn(_chain_worker, args=(world, _free_port(), n_voices, n_samples, window, result), nprocs=world, join=True)
    got = np.load(result)
    d.configure(48000)
    whole = descriptor.extract(d.Sum.many([d.Osc(f) for f in _chain_voice_freqs(n_voices)]))
    want = oracle.render(whole.words, n_samples)
    assert got.shape == want.shape and np.array_equal(got, want)
    with open(result + ".posts.json") as f:
        logs = json.load(f)
    stuck = _replay_under_stream_order(logs)
    assert not stuck, "the posted order deadlocks under one-stream-per-communicator rules: %r" % (stuck,)


def _replay_under_stream_order(logs):
    """gloo matches transfers by tag; RCCL does not.  There every transfer of one communicator is queued on ONE stream per rank,
    a send or receive occupies that stream until its counterpart on the peer has reached the head of ITS stream, and whatever
    the host waited for before posting (the comm stream's waits, the render in between) stands in front as well.  Replays the
    order the ranks posted in (logs[rank]: ("post", id, kind, peer, communicator) / ("wait", id)) under those rules; returns
    the transfers that can never start ([] = no deadlock)."""
    ops = {}      # (rank, id) -> dict
    queues = {}   # (rank, communicator) -> [ids in posting order]
    for rank, log in enumerate(logs):
        waited = []
        for entry in log:
            if entry[0] == "wait":
                waited.append(entry[1])
                continue
            _, ident, kind, peer, comm = entry
            ops[(rank, ident)] = {"kind": kind, "peer": peer, "comm": comm, "after": list(waited), "done": False}
            queues.setdefault((rank, comm), []).append(ident)
    def head(rank, comm):
        for ident in queues.get((rank, comm), []):
            if not ops[(rank, ident)]["done"]:
                return ident
        return None
    def ready(rank, ident):
        o = ops[(rank, ident)]
        return ident == head(rank, o["comm"]) and all(ops[(rank, a)]["done"] for a in o["after"])
    progress = True
    while progress:
        progress = False
        for (rank, ident), o in ops.items():
            if o["done"] or o["kind"] != "send" or not ready(rank, ident):
                continue
            other = head(o["peer"], o["comm"])
            if other is None:
                continue
            q = ops[(o["peer"], other)]
            if q["kind"] == "recv" and q["peer"] == rank and ready(o["peer"], other):
                o["done"] = q["done"] = True
                progress = True
    return [(rank, ident, o["kind"], o["peer"], o["comm"]) for (rank, ident), o in ops.items() if not o["done"]]


def test_the_replay_finds_the_deadlock_of_a_shared_communicator():
    """The order round 3's chain_mixdown posted in at two ranks (both directions on one communicator: rank 0 queues its sends S0 S1 S2 ..
    and the return receives behind them, rank 1 queues R0 R1 S'0 R2 ..) must be reported — the replay is only worth something if it is."""
    n = 4
    rank0 = [("post", k, "send", 1, "default") for k in range(n)] + [("post", n + k, "recv", 1, "default") for k in range(n)]
    rank1, ident = [], 0
    def post(kind):
        nonlocal ident
        rank1.append(("post", ident, kind, 0, "default"))
        ident += 1
        return ident - 1
    recvs = {0: post("recv")}
    for k in range(n):
        if k + 1 < n:
            recvs[k + 1] = post("recv")
        rank1.append(("wait", recvs[k]))
        post("send")
    assert _replay_under_stream_order([rank0, rank1])
    # ... and with the return path on a communicator of its own it goes through
    fixed0 = [("post", n + k, "recv", 1, "return") for k in range(n)] + [("post", k, "send", 1, "default") for k in range(n)]
    fixed1 = [e if e[0] == "wait" or e[2] == "recv" else e[:4] + ("return",) for e in rank1]
    assert not _replay_under_stream_order([fixed0, fixed1])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["osc", "gain", "ramp"])
def test_a_chain_split_between_two_programs_equals_the_whole_chain_on_the_gpu(oracle, kind):
    """dusp_render_chain_window: the first voices of a `Sum.many` in one program, the rest in another that continues the first one's raw
    sums window by window — the HIP kernels' side of shard.chain_mixdown — against the whole chain in one program and the oracle."""
    import dusp_amd as d
    from dusp_amd import descriptor, runtime
    d.configure(48000)
    n_voices, cut, n_samples, window = 11, 4, 2048 * 5 + 700, 4096
    def voice(k):
        o = d.Osc(55.0 + 13.25 * k)
        if kind == "gain": return d.Multiply(o, 0.25 + k / 16)
        if kind == "ramp": return d.Multiply(o, d.Ramp(n_samples - 900, 1, 0).trigger())
        return o
    words = lambda ks: descriptor.extract(d.Sum.many([voice(k) for k in ks])).words
    ctx = runtime.Context(0, 48000)
    whole, first, rest = ctx.build(words(range(n_voices))), ctx.build(words(range(cut))), ctx.build(words(range(cut, n_voices)))
    assert "sumchain" in whole.shape and "sumchain" in first.shape and "sumchain" in rest.shape
    want = torch.empty((1, n_samples), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    whole.render_device(n_samples, 1, None, want.data_ptr(), stream)
    got = torch.full((1, n_samples), float("nan"), dtype=torch.float32, device="cuda")
    for a in range(0, n_samples, window):
        n = min(window, n_samples - a)
        part = torch.full((1, n), float("nan"), dtype=torch.float32, device="cuda")
        first.render_chain_window(a, n, None, True, part.data_ptr(), stream)
        piece = torch.empty((1, n), dtype=torch.float32, device="cuda")
        rest.render_chain_window(a, n, part.data_ptr(), False, piece.data_ptr(), stream)
        got[:, a:a + n] = piece
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert np.array_equal(got.cpu().numpy(), oracle.render(words(range(n_voices)), n_samples))
    with pytest.raises(runtime.DuspHipError):  # windows start on whole blocks
        first.render_chain_window(100, 2048, None, True, got.data_ptr(), stream)
    other = ctx.build(descriptor.extract(d.Filter(d.Osc(100), 500)).words)
    with pytest.raises(runtime.DuspHipError):  # only programs on the fused sum chain
        other.render_chain_window(0, 2048, None, True, got.data_ptr(), stream)
    for p in (whole, first, rest, other):
        p.close()


@pytest.mark.gpu
def test_chain_mixdown_drives_the_hip_renderer_at_one_rank():
    """shard.chain_mixdown over RCCL at world_size 1 (rank 0 is first and last link): the windows of the timeline rendered by
    dusp_render_chain_window on the current stream land in the mix exactly as one dusp_render_device call writes it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    import dusp_amd as d
    from dusp_amd import descriptor, runtime
    from dusp_amd.shard import chain_mixdown
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        d.configure(48000)
        words = descriptor.extract(d.Sum.many([d.Multiply(d.Osc(f), 0.125) for f in _chain_voice_freqs(9)])).words
        prog = runtime.Context(0, 48000).build(words)
        n = 2048 * 7 + 300
        want = torch.empty((1, n), dtype=torch.float32, device="cuda")
        prog.render_device(n, 1, None, want.data_ptr(), torch.cuda.current_stream().cuda_stream)
        calls = []

        def render_window(first, count, init, raw, out):
            calls.append((first, count, init is None, raw))
            prog.render_chain_window(first, count, None if init is None else init.data_ptr(), raw, out.data_ptr(), torch.cuda.current_stream().cuda_stream)

        mix = chain_mixdown(render_window, n, 4096, want)
        torch.cuda.synchronize()
        assert calls == [(a, min(4096, n - a), True, False) for a in range(0, n, 4096)]
        assert torch.equal(mix, want)
        prog.close()
    finally:
        dist.destroy_process_group()
