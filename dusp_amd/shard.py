"""Multi-GPU sharding of independent circuit instances.

Instances (voices / sweep points) share no state, so the path shards with NO data-path collective:
rank r renders the contiguous instance range instance_range(n, r, world) on its own GPU.  The only
exchange the north star names is an optional gather of the rendered PCM onto rank 0 (RCCL over xGMI
on the GPU box, gloo in the CPU tests).

The gather works in ROUNDS of voice tiles.  In round k every peer sends the k-th tile of its shard and
rank 0 has one receive posted per peer, all in one batch (one ncclGroup on RCCL): xGMI is point-to-point,
each peer owns its own link to the root, so the inbound rate is the sum over links instead of one link at
a time.  Receives land directly in their final place of the root's [n_instances, ...] tensor — no staging
buffer, no second copy.  A round only waits for the tile it carries, so a caller can render tile k+1 on
its render stream while round k is on the wire (render_and_gather).

Mix-down (the `Sum.many` reading of BASELINE configs[2]) across ranks, two ways:
  * reduce_mixdown(): every rank folds its own voices in the reference's left-deep order (Sum.js:18-29) into one
    partial and the partials are added onto rank 0 (an f32 sum reduction; RCCL picks the order).  The summation
    order then differs from the single left-deep chain at the rank boundaries — tolerance-level, not bit-exact.
  * chain_mixdown(): the chain itself, BIT FOR BIT.  The f32 order of the adds is part of the result, so the ranks'
    voice ranges are links of ONE chain: rank r continues, for a window of the timeline, the running sums rank r-1
    left for that window (dusp_render_chain_window) and hands its own on to rank r+1; the last rank's output is the
    mix.  Windows travel down the ranks as a pipeline — rank r works on window k while rank r+1 works on window k-1 —
    so all GPUs are busy after world-1 steps; what moves per window and link is one f32 partial per sample (point
    to point: exactly the neighbour links xGMI has), and the finished windows go from the last rank to rank 0.
"""
import torch
import torch.distributed as dist


def instance_range(n_instances, rank, world):
    """Contiguous, balanced split: the first (n % world) ranks get one extra instance."""
    base, extra = divmod(n_instances, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def n_tile_rounds(n_instances, world, tile):
    """Rounds a tiled gather takes: tiles of the largest shard."""
    largest = instance_range(n_instances, 0, world)[1]
    return (largest + tile - 1) // tile


class TileGather:
    """Gathers per-rank PCM [n_local, ...] onto rank 0, one voice tile per rank and round, all peers at once.

    rank 0:  `full` is the [n_instances, ...] destination (allocated here unless handed in); round(k)
             posts one irecv per peer straight into full[peer range].
    peers:   round(k) posts one isend of local[k*tile : (k+1)*tile].
    round() returns the posted requests; wait() blocks (on RCCL: makes the current stream wait) for all of them.
    """

    def __init__(self, n_instances, local, group=None, tile=64, full=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_instances, self.tile, self.local = n_instances, max(1, int(tile)), local
        self.spans = [instance_range(n_instances, r, self.world) for r in range(self.world)]
        lo, hi = self.spans[self.rank]
        if local.shape[0] != hi - lo:
            raise ValueError("rank %d holds %d instances, its range is [%d, %d)" % (self.rank, local.shape[0], lo, hi))
        self.n_rounds = n_tile_rounds(n_instances, self.world, self.tile)
        self.full = full
        if self.rank == 0 and self.full is None:
            self.full = torch.empty((n_instances,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        self._pending = []

    def _global_rank(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def round(self, k):
        a = k * self.tile
        ops = []
        if self.rank == 0:
            lo, hi = self.spans[0]
            b = min(a + self.tile, hi - lo)
            if a < b and self.full[lo + a:lo + b].data_ptr() != self.local[a:b].data_ptr():
                self.full[lo + a:lo + b].copy_(self.local[a:b], non_blocking=True)  # the root's own shard
            for src in range(1, self.world):
                lo, hi = self.spans[src]
                b = min(a + self.tile, hi - lo)
                if a < b:
                    ops.append(dist.P2POp(dist.irecv, self.full[lo + a:lo + b], self._global_rank(src), self.group))
        else:
            lo, hi = self.spans[self.rank]
            b = min(a + self.tile, hi - lo)
            if a < b:
                ops.append(dist.P2POp(dist.isend, self.local[a:b], self._global_rank(0), self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        self._pending.extend(reqs)
        return reqs

    def wait(self):
        for r in self._pending:
            r.wait()
        self._pending = []
        return self.full


def gather_pcm(local, n_instances, group=None, tile=64, sink=None):
    """Gather per-rank PCM [n_local, channels, samples] onto rank 0 in instance order.

    Without a sink rank 0 returns the full [n_instances, channels, samples] tensor (others return None).
    `sink(lo, hi, tensor)` is called on rank 0 for every gathered range of global instances [lo, hi), after
    the round that carried it has landed (for callers that stream the PCM on instead of keeping it)."""
    g = TileGather(n_instances, local.contiguous(), group=group, tile=tile)
    for k in range(g.n_rounds):
        g.round(k)
        if sink is not None:
            g.wait()
            if g.rank == 0:
                a = k * g.tile
                for lo, hi in g.spans:
                    b = min(a + g.tile, hi - lo)
                    if a < b:
                        sink(lo + a, lo + b, g.full[lo + a:lo + b])
    g.wait()
    return g.full if (g.rank == 0 and sink is None) else None


def render_and_gather(render_tile, local, n_instances, group=None, tile=64, full=None):
    """Render this rank's shard tile by tile and gather it onto rank 0, overlapped: tile k+1 renders on the current
    stream while round k of the gather moves on a second stream (RCCL's transfers are ordered after the render of the
    tile they carry by an event, nothing else).

    render_tile(a, b) must enqueue the render of local instances [a, b) into local[a:b] on the CURRENT stream.
    Returns the root's full tensor (None elsewhere).  On CPU tensors (gloo, tests) the same rounds run synchronously."""
    g = TileGather(n_instances, local, group=group, tile=tile, full=full)
    n_local = local.shape[0]
    on_gpu = local.is_cuda
    comm = torch.cuda.Stream(device=local.device) if on_gpu else None
    for k in range(g.n_rounds):
        a, b = k * g.tile, min((k + 1) * g.tile, n_local)
        if a < b:
            render_tile(a, b)
        if on_gpu:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                g.round(k)
        else:
            g.round(k)
    if on_gpu:
        with torch.cuda.stream(comm):
            g.wait()
        torch.cuda.current_stream(local.device).wait_stream(comm)
    else:
        g.wait()
    return g.full if g.rank == 0 else None


def reduce_mixdown(partial, group=None):
    """Sum the ranks' left-deep partial mixes [channels, samples] onto rank 0 (f32 reduction).  Returns the mix on
    rank 0, None elsewhere.  The partial is overwritten on the root."""
    dist.reduce(partial, dst=(dist.get_global_rank(group, 0) if group is not None else 0), op=dist.ReduceOp.SUM, group=group)
    return partial if dist.get_rank(group) == 0 else None


_return_groups = {}  # forward group (None: the default one) -> the group the finished windows travel back on


def _chain_groups(group, return_group):
    """The communicators of chain_mixdown, made and warmed ONCE per forward group, by every rank of it at the same point.

    The finished windows go from the last rank back to rank 0 while partial sums still travel down the ranks.  On RCCL every
    transfer of one communicator is queued on one stream per side and a send occupies that stream until its receive is
    posted, so the two directions must not share a communicator: at two ranks (the last rank IS rank 1) rank 0 would queue
    its sends S0 S1 S2 .. ahead of the return receives while rank 1 queues R0 R1 S'0 R2 .. — S'0 waits for a receive behind
    S2, S2 for R2 behind S'0.  The return path therefore has a group (communicator, stream) of its own.  Both are used
    through batch_isend_irecv only, which runs on the group's own communicator; the barriers below create those
    communicators while every rank is here — a communicator made lazily inside the pipeline would need both of its ranks
    to arrive at once, which a pipeline does not promise."""
    key = group
    if key not in _return_groups:
        if return_group is None:
            world = dist.get_world_size(group)
            ranks = [dist.get_global_rank(group, r) if group is not None else r for r in range(world)]
            return_group = dist.new_group(ranks=ranks, backend=dist.get_backend(group))
        dist.barrier(group=group)
        dist.barrier(group=return_group)
        _return_groups[key] = return_group
    return _return_groups[key]


def chain_mixdown(render_window, n_samples, window, like, group=None, return_group=None):
    """One `Sum.many` chain whose voices are dealt over the ranks in contiguous runs (rank 0 the first voices), bit for bit.

    render_window(first, n, init, raw, out) must enqueue, on the CURRENT stream, this rank's links of the chain for the
    samples [first, first + n): the sums continue from `init` (a [1, n] tensor; None on rank 0) and land in `out` ([1, n]);
    raw says the sums travel on (no `x || 0`): true on every rank but the last.  With dusp_amd.runtime that is
    Program.render_chain_window(first, n, init.data_ptr(), raw, out.data_ptr(), stream).
    `window`: samples per pipeline step (a multiple of 2048).  `like`: a tensor that fixes device and dtype (float32).
    `return_group`: a group of the same ranks for the finished windows' way back (made here on first use when None: every
    rank of the DEFAULT group must then be in this call, as torch.distributed.new_group demands).
    Returns the mix [1, n_samples] on rank 0, None elsewhere.  CPU tensors (gloo, tests) run the same steps synchronously."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if window <= 0 or window % 2048 != 0:
        raise ValueError("window must be a positive multiple of 2048 samples")
    gr = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    n_windows = (n_samples + window - 1) // window
    spans = [(k * window, min((k + 1) * window, n_samples)) for k in range(n_windows)]
    on_gpu = like.is_cuda
    new = lambda n: torch.empty((1, n), dtype=torch.float32, device=like.device)
    last = world - 1
    mix = new(n_samples) if (rank == 0 or rank == last) else None  # (the last rank assembles it; rank 0 receives it, or IS the last rank)
    back = _chain_groups(group, return_group) if world > 1 else None
    comm = torch.cuda.Stream(device=like.device) if on_gpu else None

    def post(op, buf, peer, grp):  # one transfer on the group's own communicator (see _chain_groups), queued behind the comm stream
        if on_gpu:
            with torch.cuda.stream(comm):
                return dist.batch_isend_irecv([dist.P2POp(op, buf, peer, grp)])
        return dist.batch_isend_irecv([dist.P2POp(op, buf, peer, grp)])

    def wait(reqs, then_current):
        if on_gpu:
            with torch.cuda.stream(comm):
                for r in reqs:
                    r.wait()
            if then_current:
                torch.cuda.current_stream(like.device).wait_stream(comm)
        else:
            for r in reqs:
                r.wait()

    sends = []
    # the finished windows' receives: all posted before this rank's own steps, straight into their place of the mix (they are
    # on the return group: nothing of the forward direction queues behind them)
    home = []
    if rank == 0 and last != 0:
        for a, b in spans:
            home.append(post(dist.irecv, mix[:, a:b], dist.get_global_rank(back, last), back))
    # what arrives from the rank before, two windows ahead of the render (posted early so the transfer overlaps the render before it)
    inbox = {}

    def post_recv(k):
        if rank == 0 or k >= n_windows or k in inbox:
            return
        a, b = spans[k]
        buf = new(b - a)
        inbox[k] = (buf, post(dist.irecv, buf, gr(rank - 1), group))

    post_recv(0)
    for k, (a, b) in enumerate(spans):
        post_recv(k + 1)
        init = None
        if rank > 0:
            init, reqs = inbox.pop(k)
            wait(reqs, True)
        out = mix[:, a:b] if rank == last else new(b - a)
        render_window(a, b - a, init, rank != last, out)
        if on_gpu:
            ready = torch.cuda.Event()
            ready.record()
            comm.wait_event(ready)
        if rank != last:  # the partial sums travel on
            sends.append((out, post(dist.isend, out, gr(rank + 1), group)))
        elif last != 0:   # a finished window: to rank 0, on the return group
            sends.append((out, post(dist.isend, out, dist.get_global_rank(back, 0), back)))
    for reqs in home:
        wait(reqs, True)
    for _, reqs in sends:
        wait(reqs, False)
    if on_gpu:
        torch.cuda.current_stream(like.device).wait_stream(comm)
    return mix if rank == 0 else None
