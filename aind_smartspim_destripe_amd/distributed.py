"""Multi-GPU plumbing of the destripe path: one process per GPU, z-slices shard embarrassingly.

The reference parallelises over Zarr chunks with OS processes and a queue
(``zarr_destriper.py:1138-1172``); every plane is filtered independently
(``zarr_destriper.py:319-327``, overlap ``(0, 0, 0)`` at ``:1018-1022``).  Here rank == GPU, each
rank owns a contiguous z-range aligned to the Zarr z-chunk, and the only collective is one
broadcast (root 0) of the constant blob -- filter tables and, when shading is on, the flat / dark
planes -- before the data path starts.  Nothing on the data path communicates.

Two transports:

* :class:`RankGroup` -- the product path.  RCCL straight through the C ABI (``dsx_comm_*`` in
  ``include/dsx.h``, ``librccl.so`` dlopen'ed by the engine; no torch).  The 128-byte unique id goes
  from rank 0 to the other ranks through :class:`FileRendezvous` (one node) or any channel the
  caller has; ``torchrun`` is only the launcher that sets ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE``.
* the same protocol on the HOST (``transport == "host"``): when the communicator cannot be built on some rank, every
  rank agrees (over the rendezvous) to reduce / broadcast through the rendezvous directory instead.

No torch anywhere in the product: the ``torch.distributed`` (gloo) helpers the CPU tests of the sharding logic use
live in ``tests/dist_helpers.py``.
"""

import hashlib
import os
import stat
import sys
import tempfile
import threading
import time

import numpy as np


class FileRendezvous:
    """Tiny single-node key/value rendezvous on a directory: hands the RCCL unique id from rank 0 to
    the other ranks of ONE launch.  Files are written to a temporary name and renamed, so a reader
    sees a key either whole or not at all.

    The directory is keyed by the launcher's pid (``os.getppid()``: all ranks of a ``torchrun``
    launch share it, consecutive launches do not), its start time and ``MASTER_PORT``; ``DSX_RDZV_DIR`` overrides it
    for launchers whose ranks do not share a parent.  The directory is private (mode 0700) and must belong to the
    calling user: a directory somebody else pre-created under the predictable name is refused, not used.

    ``prefix`` namespaces the keys of one :class:`RankGroup` (its generation number inside the launch): a rank that has
    already opened the next group never sees -- or loses -- keys of the one rank 0 is still tearing down.
    """

    def __init__(self, rank, world, directory=None, timeout=120.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        directory = directory or os.environ.get("DSX_RDZV_DIR")
        if directory is None:
            ppid = os.getppid()
            tag = "{}_{}".format(ppid, os.environ.get("MASTER_PORT", "0"))
            try:  # the launcher's start time: a recycled pid (and a crashed run's left-over files) gets another directory
                with open("/proc/{}/stat".format(ppid)) as f:
                    tag += "_" + f.read().rsplit(")", 1)[1].split()[19]
            except (OSError, IndexError):
                pass
            directory = os.path.join(tempfile.gettempdir(), "dsx_rdzv_" + tag)
        self.dir = directory
        self.prefix = ""
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        # lstat, not stat: a symlink planted under the predictable name must not pass for the directory it points at
        st = os.lstat(self.dir)
        if stat.S_ISLNK(st.st_mode) or not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise RuntimeError(
                "rendezvous directory {} is not a private directory of this user (owner uid {}, mode {:o}); "
                "set DSX_RDZV_DIR to a directory of your own".format(self.dir, st.st_uid, st.st_mode & 0o777))

    def _path(self, key):
        return os.path.join(self.dir, self.prefix + key)

    def put(self, key, data):
        fd, tmp = tempfile.mkstemp(dir=self.dir, prefix=".tmp_")
        with os.fdopen(fd, "wb") as f:
            f.write(data)
        os.replace(tmp, self._path(key))

    def get(self, key):
        t_end = time.time() + self.timeout
        while True:
            try:
                with open(self._path(key), "rb") as f:
                    return f.read()
            except FileNotFoundError:
                if time.time() > t_end:
                    raise TimeoutError("rendezvous key {!r} did not appear in {}".format(key, self.dir))
                time.sleep(0.005)

    def barrier(self, name):
        """All ranks arrive (host-side only; the data path never needs it)."""
        self.put("{}.{}".format(name, self.rank), b"1")
        for r in range(self.world):
            self.get("{}.{}".format(name, r))

    def cleanup(self):
        """Rank 0, after a final barrier: remove this group's keys, and the directory once it is empty (a later group
        of the same launch may already have keys in it)."""
        if self.rank != 0:
            return
        now = time.time()
        for f in os.listdir(self.dir):
            path = os.path.join(self.dir, f)
            if f.startswith(".tmp_"):
                # a temporary file a put() that died left behind (nobody renames it any more) -- but a LATER group of this
                # launch may be in the middle of a put right now: only files that have been lying around for a minute
                try:
                    if now - os.lstat(path).st_mtime < 60.0:
                        continue
                except OSError:
                    continue
            elif self.prefix and not f.startswith(self.prefix):
                continue
            try:
                os.remove(path)
            except OSError:
                pass
        try:
            os.rmdir(self.dir)
        except OSError:
            pass


class RankGroup:
    """The ranks of one job around their engines' RCCL communicator (``dsx_comm_*``).

    ``RankGroup.from_env(engine)`` reads ``RANK`` / ``WORLD_SIZE`` (set by ``torchrun``); world size 1
    needs no communicator and every collective is the identity.

    Failure model of the set-up.  Stage 1 (can this rank load RCCL at all) and the outcome of stage 2 are agreed over
    the rendezvous, so a rank that fails BEFORE the collective init takes everybody to the host transport.  Stage 2
    itself -- ``ncclCommInitRank`` -- is a collective without a time-out: if one rank fails inside it (or dies), its
    peers would block for good.  A watchdog thread therefore ends the process with exit code 14 and a message when
    the init has not returned after ``DSX_COMM_TIMEOUT`` seconds (default 300): unrecoverable, but never a silent hang.
    """

    _generation = 0  # groups this process has opened: every rank of a launch opens its groups in the same order

    def __init__(self, engine, rank, world, rendezvous=None):
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        self.rdzv = None
        self.bytes_broadcast = 0
        # DSX_FORCE_COMM=1: build the communicator even for a single rank (rehearsal of the RCCL calls on a
        # one-GPU box)
        self.active = self.world > 1 or os.environ.get("DSX_FORCE_COMM") == "1"
        # "rccl": collectives over the communicator; "host": the communicator could not be built on at least one
        # rank and ALL ranks agreed (over the rendezvous) to reduce on the host instead -- the data path has no
        # collective, so the job still measures what it says, and the bench line carries the flag and the reason
        self.transport = "rccl" if self.active else "none"
        self.comm_error = None
        self._seq = 0
        if self.active:
            self.rdzv = rendezvous or FileRendezvous(self.rank, self.world)
            RankGroup._generation += 1
            self.rdzv.prefix = "g{}.".format(RankGroup._generation)
            # stage 1, no collective: can every rank load RCCL at all?  (ncclGetUniqueId on every rank; rank 0's is
            # the one that is used.)  A rank that cannot would leave the others blocked inside ncclCommInitRank.
            uid, status = b"", b"ok"
            try:
                uid = engine.comm_unique_id()
            except Exception as e:  # noqa: BLE001 - whatever the loader raised is the reason we report
                status = ("failed: {}: {}".format(type(e).__name__, e)).encode()
            bad = self._agree("preflight", status)
            # stage 2: the collective init
            if not bad:
                try:
                    if self.rank == 0:
                        self.rdzv.put("rccl_unique_id", uid)
                    else:
                        uid = self.rdzv.get("rccl_unique_id")
                    with _Watchdog(float(os.environ.get("DSX_COMM_TIMEOUT", "300")), self.rank):
                        engine.comm_init(uid, self.rank, self.world)  # returns once every rank has joined
                except Exception as e:  # noqa: BLE001
                    status = ("failed: {}: {}".format(type(e).__name__, e)).encode()
                bad = self._agree("comm_status", status)
                if bad and status == b"ok":
                    engine.comm_destroy()
            if bad:
                self.transport = "host"
                self.comm_error = "rank {}: {}".format(*bad[0])

    def _agree(self, key, status):
        """Every rank publishes its status under ``key``; returns [(rank, text)] of the ranks that are not ok."""
        self.rdzv.put("{}.{}".format(key, self.rank), status)
        statuses = [self.rdzv.get("{}.{}".format(key, r)) for r in range(self.world)]
        return [(r, st.decode(errors="replace")) for r, st in enumerate(statuses) if st != b"ok"]

    @classmethod
    def from_env(cls, engine, rendezvous=None):
        return cls(engine, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), rendezvous)

    def _host_allreduce(self, values, op):
        """Reduction through the rendezvous directory (transport "host"): every rank publishes its values under
        the call's sequence number and folds everybody's."""
        self._seq += 1
        mine = np.asarray(values, dtype=np.float64)
        self.rdzv.put("ar.{}.{}".format(self._seq, self.rank), mine.tobytes())
        rows = [np.frombuffer(self.rdzv.get("ar.{}.{}".format(self._seq, r)), dtype=np.float64)
                for r in range(self.world)]
        fold = {"sum": np.sum, "max": np.max, "min": np.min}[op]
        return [float(v) for v in fold(np.stack(rows), axis=0)]

    def allreduce(self, values, op="sum"):
        if not self.active:
            return [float(v) for v in values]
        if self.transport == "host":
            return self._host_allreduce(values, op)
        return self.engine.comm_allreduce(values, op)

    def barrier(self):
        self.allreduce([0.0])

    def broadcast_device(self, d_ptr, nbytes, root=0):
        if self.transport == "host":
            raise RuntimeError("no RCCL communicator ({}): device broadcasts are not available".format(self.comm_error))
        if self.active:
            self.engine.comm_broadcast(d_ptr, nbytes, root)
            self.bytes_broadcast += int(nbytes)

    def broadcast_constants(self, root=0):
        """RCCL broadcast of the engine's constant blob (twiddles + per-level gain tables of both configs).

        Every rank plans the same blob locally, which makes the broadcast checkable: non-root ranks
        first overwrite theirs with 0xA5 bytes, receive the root's, and the received bytes must hash to
        what the rank planned itself.  Raises ``RuntimeError`` on a mismatch -- a job whose collective
        does not work must not print a healthy result.  Returns the number of bytes broadcast."""
        ptr, nbytes = self.engine.constants_device()
        if not self.active or self.transport == "host":  # host transport: every rank keeps the blob it planned itself
            return 0
        stage = self.engine.alloc(nbytes)
        try:
            lib, ctx = self.engine._lib, self.engine._ctx
            import ctypes

            lib.dsx_memcpy_d2d(ctx, ctypes.c_void_p(stage.ptr), ctypes.c_void_p(ptr), nbytes)
            self.engine.sync()
            planned = hashlib.sha256(stage.download((nbytes,), np.uint8).tobytes()).hexdigest()
            if self.rank != root:
                stage.upload(np.full(nbytes, 0xA5, np.uint8))
            self.broadcast_device(stage, nbytes, root)
            got = stage.download((nbytes,), np.uint8)
            mine_ok = hashlib.sha256(got.tobytes()).hexdigest() == planned
            if mine_ok and self.rank != root:
                lib.dsx_memcpy_d2d(ctx, ctypes.c_void_p(ptr), ctypes.c_void_p(stage.ptr), nbytes)
                self.engine.sync()
        finally:
            stage.free()
        # the verdict is reduced BEFORE anybody raises: a rank that left early would hang the others
        all_ok = self.allreduce([1.0 if mine_ok else 0.0], "min")[0] == 1.0
        if not all_ok:
            raise RuntimeError(
                "rank {}: the broadcast constant blob differs from the locally planned one{}".format(
                    self.rank, "" if not mine_ok else " on another rank"))
        return nbytes

    def broadcast_json(self, obj, root=0):
        """Small host-side metadata (shapes, a status, a tile config) from ``root`` to every rank, as JSON through the
        rendezvous directory -- whatever the transport: it is a few hundred bytes and never on the data path."""
        if not self.active:
            return obj
        import json

        self._seq += 1
        key = "bj.{}".format(self._seq)
        if self.rank == root:
            self.rdzv.put(key, json.dumps(obj).encode())
            return obj
        return json.loads(self.rdzv.get(key).decode())

    def broadcast_array(self, array, dtype, shape, root=0):
        """A NumPy array held by ``root`` (``array`` is ignored elsewhere) -> a host copy on every rank.

        Transport "rccl": the bytes cross the node as ONE RCCL broadcast between device buffers (xGMI); transport
        "host": through one file of the rendezvous directory.  ``dtype`` / ``shape`` must agree on all ranks
        (:meth:`broadcast_json` carries them)."""
        dtype, shape = np.dtype(dtype), tuple(int(n) for n in shape)
        if not self.active:
            return np.ascontiguousarray(array, dtype=dtype).reshape(shape)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if self.transport == "host":
            self._seq += 1
            key = "ba.{}".format(self._seq)
            if self.rank == root:
                a = np.ascontiguousarray(array, dtype=dtype).reshape(shape)
                self.rdzv.put(key, a.tobytes())
                return a
            return np.frombuffer(self.rdzv.get(key), dtype=dtype).reshape(shape).copy()
        d = self.engine.alloc(max(nbytes, 16))
        try:
            if self.rank == root:
                a = np.ascontiguousarray(array, dtype=dtype).reshape(shape)
                d.upload(a)
            self.broadcast_device(d, nbytes, root)
            return a if self.rank == root else d.download(shape, dtype)
        finally:
            d.free()

    def broadcast_shading(self, flatfield, darkfield, shape_flat, shape_dark, root=0):
        """Rank ``root`` holds the flat / dark planes of a tile; every rank gets device copies
        (``dsx_set_shading_device``).  Returns the two DeviceBuffers (caller frees them)."""
        bufs = []
        for arr, shape in ((flatfield, shape_flat), (darkfield, shape_dark)):
            n = int(np.prod(shape)) * 4
            d = self.engine.alloc(n)
            if self.rank == root:
                d.upload(np.ascontiguousarray(arr, dtype=np.float32).reshape(shape))
            self.broadcast_device(d, n, root)
            bufs.append(d)
        return bufs

    def close(self):
        if self.active:
            try:
                self.barrier()
                # rank 0 removes the rendezvous directory: only once every other rank has said it is done reading
                # (with the host transport the barrier itself lives in those files)
                if self.rdzv is not None:
                    if self.rank != 0:
                        self.rdzv.put("bye.{}".format(self.rank), b"1")
                    else:
                        for r in range(1, self.world):
                            self.rdzv.get("bye.{}".format(r))
            finally:
                if self.transport == "rccl":
                    self.engine.comm_destroy()
                if self.rdzv is not None:
                    self.rdzv.cleanup()


class _Watchdog:
    """Ends the process (exit code 14) if the guarded block has not finished after ``seconds``: the collective
    communicator init has no time-out of its own, and a peer that failed inside it never arrives."""

    def __init__(self, seconds, rank):
        self.seconds, self.rank = seconds, rank
        self.timer = None

    def _fire(self):
        sys.stderr.write("[dsx] rank {}: the RCCL communicator init did not return within {:.0f} s (a peer failed "
                         "inside it or died); giving up\n".format(self.rank, self.seconds))
        sys.stderr.flush()
        os._exit(14)

    def __enter__(self):
        if self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def z_shard(n_slices, world_size, rank, z_chunk=64):
    """Contiguous ``[start, stop)`` z-range of ``rank``: whole z-chunks, spread as evenly as possible.

    Chunk alignment keeps every output Zarr chunk ``(1, 1, 64, 128, 128)``
    (``zarr_destriper.py:1066-1074``) written by exactly one rank, so no locking is needed -- the same
    argument that makes the reference's concurrent consumers safe.
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    n_chunks = (n_slices + z_chunk - 1) // z_chunk
    base, extra = divmod(n_chunks, world_size)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    start = min(first * z_chunk, n_slices)
    stop = min((first + count) * z_chunk, n_slices)
    return start, stop
