"""Channel sharding across ranks, covered on CPU with gloo at world_size 2 (the N>1 path of
bench.py uses the same helpers with backend nccl = RCCL)."""

import os
import socket

import numpy as np
import pytest

from sdft_amd import shard


def test_channel_blocks_partition():
    for channels in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 4, 8):
            blocks = shard.all_blocks(channels, world)
            assert blocks[0][0] == 0
            assert sum(c for _, c in blocks) == channels
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            sizes = [c for _, c in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard.weak_scaling_channels(64, 8) == 512
    with pytest.raises(ValueError):
        shard.channel_block(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import oracle as O
    from sdft_amd.signals import sine_sweep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        channels, n, m = 5, 600, 32
        first, count = shard.channel_block(channels, world, rank)
        # every rank analyses only its own channels (the oracle stands in for the device here:
        # this test covers the partition + reduction logic, not the kernels)
        local = 0.0
        for c in range(first, first + count):
            x = sine_sweep(n, channel=c, channels=channels)
            dig, _ = O.Port(m).digest(x, with_y=False)
            local += float(dig[:, 2].sum())
        shard.barrier()
        total = shard.sum_over_ranks(local)
        rate, secs = shard.job_throughput(count * n, 0.5 + rank)      # slowest rank defines the time
        census = shard.run_census(0.5 + rank, 0.7 - 0.1 * rank, rank)     # (round 4) what makes a multi-GPU record checkable
        q.put((rank, first, count, total, rate, secs, census))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from sdft_amd.signals import sine_sweep
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    census = [r[-1] for r in res]
    res = [r[:-1] for r in res]
    for cs in census:                                              # the same census on every rank
        assert cs["backend"] == "gloo" and cs["ranks"] == 2 and cs["world_size"] == 2 and cs["distinct_local_devices"] == 2
        assert cs["seconds_min"] == 0.5 and cs["seconds_max"] == 1.5
        assert np.isclose(cs["roofline_frac_min"], 0.6) and np.isclose(cs["roofline_frac_max"], 0.7)
    assert shard.run_census(1.0, 0.5) == {"backend": None, "ranks": 1, "world_size": 1}     # no process group: one rank
    (r0, f0, c0, t0, rate0, s0), (r1, f1, c1, t1, rate1, s1) = res
    assert (f0, c0, f1, c1) == (0, 3, 3, 2)
    want = 0.0
    for c in range(5):
        dig, _ = O.Port(32).digest(sine_sweep(600, channel=c, channels=5), with_y=False)
        want += float(dig[:, 2].sum())
    assert np.isclose(t0, want, rtol=1e-12) and t0 == t1          # same all-reduced value on both ranks
    assert s0 == s1 == 1.5 and np.isclose(rate0, 5 * 600 / 1.5) and rate0 == rate1
