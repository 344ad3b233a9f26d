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
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist

    import dist_helpers as dh
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
        flat, dark = dh.broadcast_shading(dist, flat0 if rank == 0 else None, dark0 if rank == 0 else None,
                                          (32, 48), (40, 50))  # fmt: skip
        ok_bcast = bool(np.array_equal(flat, flat0) and np.array_equal(dark, dark0))
        # z-sharding of a 200-slice stack with 64-slice chunks; each rank "processes" its own planes
        n = 200
        s, e = dd.z_shard(n, world, rank, 64)
        checksum = 0
        for z in range(s, e):
            checksum += int(synth.synthetic_plane(z % 4, 16, 16).astype(np.uint64).sum()) * (z + 1)
        total, tmax = dh.reduce_counters(dist, e - s, 1.0 + rank)
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


# ---- RankGroup (RCCL through the C ABI): rendezvous + protocol, with the engine's collectives stubbed ----
class _FakeBuf:
    def __init__(self, eng, n):
        self.eng, self.nbytes, self.data, self.ptr = eng, n, np.zeros(n, np.uint8), id(self)

    def upload(self, a):
        self.data[: a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)

    def download(self, shape, dtype):
        return self.data[: int(np.prod(shape)) * np.dtype(dtype).itemsize].view(dtype).reshape(shape).copy()

    def free(self):
        pass


class _FakeEngine:
    """Stands in for DestripeEngine on a machine without GPUs: the comm_* calls exchange through files in
    the rendezvous directory, so the RankGroup protocol (id hand-off, blob poisoning + hash check,
    min-reduce of the verdict) runs end to end in real processes."""

    def __init__(self, rank, world, xdir, corrupt=False):
        self.rank, self.world, self.xdir, self.corrupt, self.n = rank, world, xdir, corrupt, 0
        self.blob = _FakeBuf(self, 4096)
        self.blob.data[:] = np.arange(4096) % 251
        self._lib, self._ctx = self, None

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        assert uid == bytes(range(128)) and (rank, world) == (self.rank, self.world)

    def comm_destroy(self):
        pass

    def constants_device(self):
        return self.blob, self.blob.nbytes

    def alloc(self, n):
        return _FakeBuf(self, n)

    def sync(self):
        pass

    def dsx_memcpy_d2d(self, ctx, dst, src, n):  # ctypes.c_void_p(buffer) is not needed for the fake
        raise AssertionError("patched below")

    def _xchg(self, tag, payload=None):
        from aind_smartspim_destripe_amd.distributed import FileRendezvous

        r = FileRendezvous(self.rank, self.world, self.xdir)
        self.n += 1
        if payload is not None:
            r.put("{}{}.{}".format(tag, self.n, self.rank), payload)
        return r

    def comm_broadcast(self, buf, nbytes, root):
        r = self._xchg("b", buf.data.tobytes() if self.rank == root else None)
        got = np.frombuffer(r.get("b{}.{}".format(self.n, root)), np.uint8).copy()
        if self.corrupt and self.rank != root:
            got[7] ^= 1
        buf.data[:nbytes] = got[:nbytes]

    def comm_allreduce(self, values, op):
        r = self._xchg("a", np.asarray(values, np.float64).tobytes())
        allv = np.stack([np.frombuffer(r.get("a{}.{}".format(self.n, k)), np.float64) for k in range(self.world)])
        return list({"sum": allv.sum(0), "max": allv.max(0), "min": allv.min(0)}[op])


def _rankgroup_worker(rank, world, xdir, corrupt, q):
    sys.path.insert(0, REPO)
    import ctypes

    from aind_smartspim_destripe_amd import distributed as dd

    eng = _FakeEngine(rank, world, xdir, corrupt)
    real_vp = ctypes.c_void_p
    bufs = {}

    def fake_vp(x):  # RankGroup wraps addresses in c_void_p; keep the buffer objects reachable by "address"
        bufs[id(x) if not isinstance(x, int) else x] = x
        return x

    def d2d(ctx, dst, src, n):
        d = dst if isinstance(dst, _FakeBuf) else next(b for b in (eng.blob,) if b.ptr == dst or b is dst)
        s = src if isinstance(src, _FakeBuf) else next(b for b in (eng.blob,) if b.ptr == src or b is src)
        d.data[:n] = s.data[:n]

    eng.dsx_memcpy_d2d = d2d
    ctypes.c_void_p = fake_vp
    try:
        # _FakeBuf.ptr must round-trip through the fake c_void_p: use the object itself as its address
        eng.blob.ptr = eng.blob
        orig_alloc = eng.alloc

        def alloc(n):
            b = orig_alloc(n)
            b.ptr = b
            return b

        eng.alloc = alloc
        grp = dd.RankGroup(eng, rank, world, dd.FileRendezvous(rank, world, xdir))
        try:
            n = grp.broadcast_constants(root=0)
            tot = grp.allreduce([float(rank + 1)], "sum")[0]
            # metadata over the rendezvous, a plane over the (fake) device broadcast: only rank 0 holds them
            meta = grp.broadcast_json({"shape": [5, 7], "who": "rank0"} if rank == 0 else None, root=0)
            plane0 = (np.arange(35, dtype=np.float32).reshape(5, 7) * 0.5) if rank == 0 else None
            plane = grp.broadcast_array(plane0, np.float32, meta["shape"], root=0)
            assert meta == {"shape": [5, 7], "who": "rank0"}
            if not corrupt:
                assert np.array_equal(plane, np.arange(35, dtype=np.float32).reshape(5, 7) * 0.5)
            q.put((rank, "ok", n, tot))
        except RuntimeError as e:
            q.put((rank, "error", str(e), 0.0))
    finally:
        ctypes.c_void_p = real_vp


@pytest.mark.parametrize("corrupt", [False, True])
def test_rankgroup_protocol_three_ranks(tmp_path, corrupt):
    """Unique-id hand-off through the file rendezvous, constants broadcast with the hash check, and the
    verdict min-reduced over the ranks: a corrupted broadcast must raise on EVERY rank (fail loudly)."""
    import multiprocessing as mp

    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rankgroup_worker, args=(r, world, str(tmp_path), corrupt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if corrupt:
        assert all(r[1] == "error" for r in res), res
    else:
        assert [r[1] for r in res] == ["ok"] * world and all(r[2] == 4096 and r[3] == 6.0 for r in res), res


def _host_transport_worker(rank, world, xdir, mode, q):
    sys.path.insert(0, REPO)
    from aind_smartspim_destripe_amd import distributed as dd

    eng = _FakeEngine(rank, world, xdir)
    destroyed = []
    eng.comm_destroy = lambda: destroyed.append(rank)
    if rank == 1 and mode == "load":  # this rank cannot load RCCL at all: caught before anybody enters the init
        def no_lib():
            raise OSError("librccl.so: cannot open shared object file")

        eng.comm_unique_id = no_lib
        eng.comm_init = lambda *a: (_ for _ in ()).throw(AssertionError("comm_init after a failed preflight"))
    elif mode == "load":
        eng.comm_init = lambda *a: (_ for _ in ()).throw(AssertionError("comm_init after a failed preflight"))
    elif rank == 1:  # the collective init itself fails on this rank
        def broken(uid, rank_, world_):
            raise OSError("librccl.so: ncclCommInitRank: unhandled system error")

        eng.comm_init = broken
    grp = dd.RankGroup(eng, rank, world, dd.FileRendezvous(rank, world, xdir))
    n = grp.broadcast_constants(root=0)
    mx = grp.allreduce([float(rank), 10.0 - rank], "max")
    mn = grp.allreduce([float(rank)], "min")
    # arrays still reach every rank on the host transport (through the rendezvous directory)
    arr = grp.broadcast_array(np.arange(6, dtype=np.uint16).reshape(2, 3) if rank == 0 else None, np.uint16, (2, 3))
    assert np.array_equal(arr, np.arange(6, dtype=np.uint16).reshape(2, 3))
    assert grp.broadcast_json([1, {"a": None}] if rank == 0 else "ignored") == [1, {"a": None}]
    grp.barrier()
    try:
        grp.broadcast_device(None, 16, 0)
        bcast = "no error"
    except RuntimeError:
        bcast = "raises"
    import time

    if rank == 2:
        time.sleep(0.3)  # a late reader: rank 0 must not remove the rendezvous files under it
    grp.close()
    q.put((rank, grp.transport, grp.comm_error, n, mx, mn, bcast, list(destroyed)))


@pytest.mark.parametrize("mode", ["load", "init"])
def test_rankgroup_agrees_on_host_transport_when_rccl_is_missing(tmp_path, mode):
    """One rank cannot build the RCCL communicator: EVERY rank must learn it (over the rendezvous), drop its own
    communicator and reduce on the host -- same results, flagged transport, device broadcasts refused."""
    import multiprocessing as mp

    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_host_transport_worker, args=(r, world, str(tmp_path), mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, transport, err, n, mx, mn, bcast, destroyed in res:
        assert transport == "host" and "rank 1" in err and "librccl" in err
        assert n == 0 and mx == [2.0, 10.0] and mn == [0.0] and bcast == "raises"
        # ranks that had joined release their communicator; after a failed preflight nobody joined
        assert destroyed == ([] if (rank == 1 or mode == "load") else [rank])


def test_file_rendezvous_atomic_and_timeout(tmp_path):
    from aind_smartspim_destripe_amd.distributed import FileRendezvous

    a = FileRendezvous(0, 2, str(tmp_path), timeout=0.2)
    b = FileRendezvous(1, 2, str(tmp_path), timeout=0.2)
    with pytest.raises(TimeoutError):
        b.get("missing")
    a.put("k", b"x" * 128)
    assert b.get("k") == b"x" * 128
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith(".tmp_")]
    a.cleanup()
    assert not os.path.exists(str(tmp_path)) or not os.listdir(str(tmp_path))


def test_file_rendezvous_refuses_a_directory_it_does_not_own_privately(tmp_path):
    """The default directory name is predictable and lives in the world-writable temp dir: a directory somebody
    else could write to (here: group / other bits set) must be refused, not trusted."""
    from aind_smartspim_destripe_amd.distributed import FileRendezvous

    d = tmp_path / "loose"
    d.mkdir()
    os.chmod(str(d), 0o777)
    with pytest.raises(RuntimeError, match="not a private directory"):
        FileRendezvous(0, 2, str(d))
    os.chmod(str(d), 0o700)
    FileRendezvous(0, 2, str(d))  # fine once private
    fresh = tmp_path / "made_by_us"
    FileRendezvous(0, 2, str(fresh))
    assert (os.stat(str(fresh)).st_mode & 0o777) == 0o700


def _two_groups_worker(rank, world, xdir, q):
    sys.path.insert(0, REPO)
    import time

    from aind_smartspim_destripe_amd import distributed as dd

    out = []
    for gen in range(2):
        eng = _FakeEngine(rank, world, xdir)
        eng.comm_init = lambda *a: (_ for _ in ()).throw(OSError("no RCCL here"))  # -> host transport: all in files
        grp = dd.RankGroup(eng, rank, world, dd.FileRendezvous(rank, world, xdir, timeout=20.0))
        out.append(grp.allreduce([float(rank + 10 * gen)], "sum")[0])
        if rank == 0 and gen == 0:
            time.sleep(0.5)  # rank 0 is slow to tear group 1 down: rank 1 is already inside group 2 by then
        grp.close()
    q.put((rank, out))


def test_consecutive_rank_groups_of_one_launch_do_not_touch_each_others_keys(tmp_path):
    """Keys are namespaced per group generation: a fast rank that has opened the next RankGroup keeps its fresh keys
    while rank 0 is still removing the previous group's (they used to share names: a deleted 'preflight' key meant a
    120 s time-out, a stale unique id a wrong communicator)."""
    import multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_groups_worker, args=(r, world, str(tmp_path / "rdzv"), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, [1.0, 21.0]), (1, [1.0, 21.0])], res


def test_comm_init_watchdog_ends_a_rank_whose_peer_never_arrives(tmp_path):
    """ncclCommInitRank has no time-out: the watchdog turns 'blocked for good' into exit code 14 and a message."""
    import subprocess

    code = (
        "import sys, time; sys.path.insert(0, {!r})\n"
        "from aind_smartspim_destripe_amd.distributed import _Watchdog\n"
        "with _Watchdog(0.3, 5):\n"
        "    time.sleep(30)\n"
    ).format(REPO)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 14 and "rank 5" in r.stderr and "did not return" in r.stderr


# ---- destripe_zarr under two ranks on ONE store (engine replaced by the CPU oracle) ---------------------
class _RdzvBarrier:
    def __init__(self, rank, world, directory):
        from aind_smartspim_destripe_amd.distributed import FileRendezvous

        self.r, self.n = FileRendezvous(rank, world, directory), 0

    def barrier(self):
        self.n += 1
        self.r.barrier("b{}".format(self.n))


def _oracle_destripe_planes(planes, input_tile_path, no_cells_config, cells_config, shadow_correction=None,
                            microscope_high_int=2700, out_dtype=np.uint16, **_):
    from oracle import destripe_oracle as orc

    out = [orc.filter_stripes(p, input_tile_path, no_cells_config, cells_config, shadow_correction, microscope_high_int)
           for p in planes]  # fmt: skip
    return np.clip(np.stack(out), 0, 65535).astype(out_dtype)


def _zarr_rank_worker(rank, world, src_path, out_path, rdzv_dir, q):
    sys.path.insert(0, REPO)
    from aind_smartspim_destripe_amd import filtering as fl
    from aind_smartspim_destripe_amd import synth, zarr_destriper as zd

    fl.destripe_planes = _oracle_destripe_planes  # no GPU here: the chunk-map logic is what is under test
    group = _RdzvBarrier(rank, world, rdzv_dir) if world > 1 else None
    n, _ = zd.destripe_zarr_store(src_path, out_path, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, None,
                            prediction_chunksize=(4, 48, 64), output_chunks=(1, 1, 4, 16, 16), rank=rank,
                            world_size=world, device=0, device_retile=False, group=group)  # fmt: skip
    # codec threads: the cores of this process shared among the ranks of the node, at least 2
    cores = len(os.sched_getaffinity(0))
    assert zd.LAST_RUN["io_threads"] == max(2, min(cores // world, 64)), (zd.LAST_RUN, cores, world)
    assert zd.LAST_RUN["rank"] == rank and zd.LAST_RUN["world_size"] == world
    q.put((rank, n))


def test_two_rank_destripe_zarr_matches_single_rank(tmp_path):
    """Both ranks run destripe_zarr on one store (z-ranges of whole output chunks, rank 0 creates the array
    anew and atomically, a stale array of another geometry is ignored); the result equals a one-rank run."""
    import multiprocessing as mp

    from aind_smartspim_destripe_amd import synth
    from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

    vol = np.stack([synth.synthetic_plane(k, 48, 64) for k in range(12)])
    src = MiniZarrArray.create(str(tmp_path / "in.zarr"), (1, 1, 12, 48, 64), (1, 1, 4, 16, 16), np.uint16)
    src[0, 0] = vol
    # stale output of an earlier run with another shape: must not be picked up by the waiting rank
    MiniZarrArray.create(str(tmp_path / "out2.zarr"), (1, 1, 3, 8, 8), (1, 1, 1, 8, 8), np.uint16)
    ctx = mp.get_context("spawn")
    results = {}
    for world, out in ((1, "out1.zarr"), (2, "out2.zarr")):
        q = ctx.Queue()
        procs = [ctx.Process(target=_zarr_rank_worker,
                             args=(r, world, str(tmp_path / "in.zarr"), str(tmp_path / out), str(tmp_path / "rdzv"), q))
                 for r in range(world)]  # fmt: skip
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=300) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = got
    assert results[1] == [(0, 12)] and results[2] == [(0, 8), (1, 4)]  # 3 output z-chunks: 2 + 1
    a = MiniZarrArray.open(str(tmp_path / "out1.zarr"))
    b = MiniZarrArray.open(str(tmp_path / "out2.zarr"))
    assert a.shape == b.shape == (1, 1, 12, 48, 64)
    np.testing.assert_array_equal(a[0, 0], b[0, 0])
    assert a[0, 0].std() > 0


def _zarr_rank_worker_nogroup(rank, world, src_path, out_path, delay, q):
    sys.path.insert(0, REPO)
    import time

    from aind_smartspim_destripe_amd import filtering as fl
    from aind_smartspim_destripe_amd import synth, zarr_destriper as zd

    fl.destripe_planes = _oracle_destripe_planes
    time.sleep(delay)
    n, _ = zd.destripe_zarr_store(src_path, out_path, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, None,
                            prediction_chunksize=(4, 48, 64), output_chunks=(1, 1, 4, 16, 16), rank=rank,
                            world_size=world, device=0, device_retile=False, group=None, compressor="blosc")  # fmt: skip
    q.put((rank, n))


def test_groupless_rank_ignores_a_stale_array_of_the_same_geometry_but_another_codec(tmp_path):
    """No barrier between the ranks: rank 1 polls the metadata.  The output directory holds the array of an earlier
    RAW run with the very same geometry; rank 1 starts first and must wait for rank 0's Blosc metadata instead of
    writing raw chunks under it (the store would be unreadable, with no error anywhere)."""
    import multiprocessing as mp

    from aind_smartspim_destripe_amd import synth
    from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

    vol = np.stack([synth.synthetic_plane(k, 48, 64) for k in range(8)])
    src = MiniZarrArray.create(str(tmp_path / "in.zarr"), (1, 1, 8, 48, 64), (1, 1, 4, 16, 16), np.uint16)
    src[0, 0] = vol
    stale = MiniZarrArray.create(str(tmp_path / "out.zarr"), (1, 1, 8, 48, 64), (1, 1, 4, 16, 16), np.uint16, compressor=None)
    assert stale.matches((1, 1, 8, 48, 64), (1, 1, 4, 16, 16), np.uint16)
    assert not stale.matches((1, 1, 8, 48, 64), (1, 1, 4, 16, 16), np.uint16, "blosc")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zarr_rank_worker_nogroup,
                         args=(r, 2, str(tmp_path / "in.zarr"), str(tmp_path / "out.zarr"), 1.5 if r == 0 else 0.0, q))
             for r in range(2)]  # fmt: skip
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, 4), (1, 4)]
    out = MiniZarrArray.open(str(tmp_path / "out.zarr"))
    assert out.compressor is not None and out.compressor[0] == "blosc"
    res = out[0, 0]  # every chunk decodes as a Blosc frame: nobody wrote raw bytes under the new metadata
    ref = _oracle_destripe_planes(vol, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, 2500)
    np.testing.assert_array_equal(res, ref)


def test_destripe_channel_needs_group_for_many_ranks(tmp_path):
    from aind_smartspim_destripe_amd import zarr_destriper as zd

    with pytest.raises(ValueError):
        zd.destripe_channel(str(tmp_path), str(tmp_path), "Ex_561_Em_593", str(tmp_path), None, {}, {}, {}, world_size=2)


# ---- destripe_channel under two ranks: rank 0 alone reads the flat and dark planes, the group broadcasts them ----
def _channel_rank_worker(rank, world, root, rdzv_dir, q):
    sys.path.insert(0, REPO)
    from aind_smartspim_destripe_amd import distributed as dd
    from aind_smartspim_destripe_amd import filtering as fl
    from aind_smartspim_destripe_amd import mini_tiff, synth, zarr_destriper as zd

    fl.destripe_planes = _oracle_destripe_planes  # no GPU here: plumbing under test
    reads = []
    real_imread = mini_tiff.imread

    def counting_imread(path, *a, **k):
        reads.append(os.path.basename(str(path)))
        if rank != 0 and world > 1:
            raise AssertionError("rank {} opened {}: only rank 0 reads the shading planes".format(rank, path))
        return real_imread(path, *a, **k)

    mini_tiff.imread = counting_imread
    group = None
    if world > 1:
        eng = _FakeEngine(rank, world, rdzv_dir)
        orig_alloc = eng.alloc
        eng.alloc = lambda n: orig_alloc(n)
        group = dd.RankGroup(eng, rank, world, dd.FileRendezvous(rank, world, rdzv_dir))
    params = {"cells_config": synth.CELLS_CONFIG, "no_cells_config": synth.NO_CELLS_CONFIG}
    d = os.path.join(root, "derivatives")
    done = zd.destripe_channel(
        zarr_dataset_path=os.path.join(root, "data"), channel_name="Ex_488_Em_525",
        results_folder=os.path.join(root, "results{}".format(world)), derivatives_path=d, xyz_resolution=[1.8, 1.8, 2.0],
        estimated_channel_flats=[os.path.join(d, "flat_0.tif"), os.path.join(d, "flat_1.tif")],
        laser_tiles={"0": ["431040_368180"], "1": ["431040_394100"]}, parameters=params,
        prediction_chunksize=(4, 32, 48), output_chunks=(1, 1, 4, 16, 16), compressor="zlib", n_levels=1,
        rank=rank, world_size=world, device=0, group=group, device_retile=False)  # fmt: skip
    nbytes = group.bytes_broadcast if group is not None else 0
    if group is not None:
        group.close()
    q.put((rank, done, sorted(set(reads)), nbytes))


def test_two_rank_destripe_channel_broadcasts_the_shading_planes(tmp_path):
    """``destripe_channel`` with the reference's keyword set under two ranks and a ``RankGroup``: rank 0 reads the flat
    of each tile's laser side and ``DarkMaster_cropped.tif`` ONCE per tile and the group hands them to rank 1 (which
    never opens a TIFF); both ranks correct with the same planes, and the store equals a one-rank run."""
    import multiprocessing as mp

    from aind_smartspim_destripe_amd import mini_tiff, synth
    from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

    H, W, Z = 32, 48, 8
    names = ["431040_368180", "431040_394100"]
    for t, name in enumerate(names):
        a = MiniZarrArray.create(str(tmp_path / "data" / "Ex_488_Em_525" / (name + ".zarr") / "0"), (1, 1, Z, H, W),
                                 (1, 1, 4, 16, 16), np.uint16, compressor="zlib")  # fmt: skip
        a[0, 0] = synth.synthetic_stack(Z, H, W, n_unique=4) + np.uint16(3 * t)
    d = tmp_path / "derivatives"
    d.mkdir()
    mini_tiff.imwrite(str(d / "DarkMaster_cropped.tif"), np.full((H + 8, W + 8), 90, np.uint16))
    yy, xx = np.mgrid[0:H, 0:W]
    for side in (0, 1):
        f = (1.0 + 0.2 * side - 0.3 * ((yy - H / 2) / H) ** 2 - 0.2 * ((xx - W / 2) / W) ** 2).astype(np.float32)
        mini_tiff.imwrite(str(d / "flat_{}.tif".format(side)), f)
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        q = ctx.Queue()
        procs = [ctx.Process(target=_channel_rank_worker, args=(r, world, str(tmp_path), str(tmp_path / "rdzv"), q))
                 for r in range(world)]  # fmt: skip
        for p in procs:
            p.start()
        results[world] = sorted(q.get(timeout=300) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    tiles = [n + ".zarr" for n in names]
    assert results[1][0][1] == {t: Z for t in tiles}
    (r0, done0, reads0, sent0), (r1, done1, reads1, sent1) = results[2]
    assert done0 == {t: 4 for t in tiles} and done1 == {t: 4 for t in tiles}  # two output z-chunks: one each
    assert reads0 == ["DarkMaster_cropped.tif", "flat_0.tif", "flat_1.tif"] and reads1 == []
    plane_bytes = H * W * 4 + (H + 8) * (W + 8) * 2
    assert sent0 == sent1 == 2 * plane_bytes  # per tile: one flat (float32) + one dark (uint16), nothing else
    for t in tiles:
        a = MiniZarrArray.open(str(tmp_path / "results1" / "destriped_data" / "Ex_488_Em_525" / t / "0"))[0, 0]
        b = MiniZarrArray.open(str(tmp_path / "results2" / "destriped_data" / "Ex_488_Em_525" / t / "0"))[0, 0]
        np.testing.assert_array_equal(a, b)
        assert a.std() > 0
