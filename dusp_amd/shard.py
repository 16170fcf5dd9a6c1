"""Multi-GPU sharding of independent circuit instances.

Instances (voices / sweep points) share no state, so the path shards with NO data-path collective:
rank r renders the contiguous instance range instance_range(n, r, world) on its own GPU.  The only
exchange the north star names is an optional gather of the rendered PCM onto rank 0 (RCCL over xGMI
on the GPU box, gloo in the CPU tests), done tile by tile so the root never has to hold more than
one tile per peer beyond its own shard.
"""
import torch
import torch.distributed as dist


def instance_range(n_instances, rank, world):
    """Contiguous, balanced split: the first (n % world) ranks get one extra instance."""
    base, extra = divmod(n_instances, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_pcm(local, n_instances, group=None, tile=64, sink=None):
    """Gather per-rank PCM [n_local, channels, samples] onto rank 0 in instance order.

    `sink(lo, hi, tensor)` is called on rank 0 for every gathered tile of global instances [lo, hi);
    without a sink rank 0 returns the full [n_instances, channels, samples] tensor (others return None).
    Point-to-point sends keep each peer on its single xGMI link to the root; tiles bound the root's
    staging memory.
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    full = None
    if rank == 0 and sink is None:
        full = torch.empty((n_instances,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)

        def sink(lo, hi, t):  # noqa: F811
            full[lo:hi] = t

    for src in range(world):
        lo, hi = instance_range(n_instances, src, world)
        for a in range(lo, hi, tile):
            b = min(a + tile, hi)
            if src == 0:
                if rank == 0:
                    sink(a, b, local[a - lo:b - lo])
            elif rank == src:
                dist.send(local[a - lo:b - lo].contiguous(), dst=0, group=group)
            elif rank == 0:
                buf = torch.empty((b - a,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                dist.recv(buf, src=src, group=group)
                sink(a, b, buf)
    return full
