"""N > 1 path on CPU: world_size-2 gloo process group (z-sharding, constant broadcast, counters)."""

import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from aind_smartspim_destripe_amd import distributed as dsx_dist  # noqa: E402


def test_z_shard_covers_and_aligns():
    for n, world, chunk in [(4096, 8, 64), (4096, 3, 64), (100, 4, 64), (64, 8, 64), (1000, 7, 64), (5, 2, 64)]:
        ranges = [dsx_dist.z_shard(n, world, r, chunk) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0 and a0 <= a1
        for s, e in ranges:
            assert s % chunk == 0 or s == n
        chunks = [-(-(e - s) // chunk) for s, e in ranges]  # whole or (last) partial chunks per rank
        assert max(chunks) - min(chunks) <= 1
    assert dsx_dist.z_shard(4096, 8, 3) == (1536, 2048)  # 512 slices = 8 z-chunks per GPU
    with pytest.raises(ValueError):
        dsx_dist.z_shard(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist

    from aind_smartspim_destripe_amd import distributed as dd
    from aind_smartspim_destripe_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # constants / shading broadcast: only rank 0 has the real planes
        rs = np.random.RandomState(3)
        flat0 = (0.5 + rs.rand(32, 48)).astype(np.float32)
        dark0 = (100 * rs.rand(40, 50)).astype(np.float32)
        flat, dark = dd.broadcast_shading(dist, flat0 if rank == 0 else None, dark0 if rank == 0 else None,
                                          (32, 48), (40, 50))  # fmt: skip
        ok_bcast = bool(np.array_equal(flat, flat0) and np.array_equal(dark, dark0))
        # z-sharding of a 200-slice stack with 64-slice chunks; each rank "processes" its own planes
        n = 200
        s, e = dd.z_shard(n, world, rank, 64)
        checksum = 0
        for z in range(s, e):
            checksum += int(synth.synthetic_plane(z % 4, 16, 16).astype(np.uint64).sum()) * (z + 1)
        total, tmax = dd.reduce_counters(dist, e - s, 1.0 + rank)
        q.put((rank, s, e, checksum, ok_bcast, total, tmax))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_broadcast():
    import torch.multiprocessing as mp

    from aind_smartspim_destripe_amd import synth

    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, e0, c0, b0, t0, m0), (r1, s1, e1, c1, b1, t1, m1) = res
    assert (s0, e0, s1, e1) == (0, 128, 128, 200)
    assert b0 and b1
    assert t0 == t1 == 200.0 and m0 == m1 == 2.0
    expect = sum(int(synth.synthetic_plane(z % 4, 16, 16).astype(np.uint64).sum()) * (z + 1) for z in range(200))
    assert c0 + c1 == expect
