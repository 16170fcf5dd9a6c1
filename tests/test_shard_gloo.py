"""N>1 path on CPU: two gloo ranks shard the voices, each 'renders' its own range, rank 0 gathers.

The GPU box runs the same host logic with backend 'nccl' (RCCL); here the renderer is the CPU oracle
(test infrastructure) so that the gathered PCM can be checked against a single-process render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from dusp_amd.shard import instance_range


def test_instance_range_partitions_exactly():
    for n in (1, 7, 64, 1000, 65536):
        for world in (1, 2, 3, 4, 8):
            spans = [instance_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_voices, n_samples, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dusp_amd as d
    from dusp_amd import descriptor
    from dusp_amd.shard import gather_pcm, instance_range
    from oracle import oracle
    d.configure(48000)
    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Osc(20 + k / 8), d.Ramp(n_samples, 1, 0).trigger()))
                            for k in range(n_voices)])
    lo, hi = instance_range(n_voices, rank, world)
    local = np.stack([oracle.render(uni.words, n_samples, params=uni.params, n_instances=n_voices, instance=i)
                      for i in range(lo, hi)])
    full = gather_pcm(torch.from_numpy(local), n_voices, tile=2)
    if rank == 0:
        np.save(result_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path, oracle):
    n_voices, n_samples, world = 7, 700, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, port, n_voices, n_samples, result), nprocs=world, join=True)
    got = np.load(result)
    import dusp_amd as d
    from dusp_amd import descriptor
    d.configure(48000)
    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Osc(20 + k / 8), d.Ramp(n_samples, 1, 0).trigger()))
                            for k in range(n_voices)])
    want = np.stack([oracle.render(uni.words, n_samples, params=uni.params, n_instances=n_voices, instance=i)
                     for i in range(n_voices)])
    assert got.shape == want.shape and np.array_equal(got, want)
