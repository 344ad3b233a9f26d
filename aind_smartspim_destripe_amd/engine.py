"""ctypes binding of ``libdsx_hip.so`` (C ABI in ``include/dsx.h``).

No GPU array library is involved: device memory, copies, streams and timing all go through the
C ABI.  The engine has NO CPU fallback -- if the library is missing or no MI355X is visible the
constructor raises :class:`DsxError`.
"""

import ctypes
import os

import numpy as np

from . import wavelets

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSX_LIB overrides the library path (A/B runs of two builds inside one GPU session)
LIB_PATH = os.environ.get("DSX_LIB") or os.path.join(_HERE, "_lib", "libdsx_hip.so")

DSX_U16, DSX_F32 = 0, 1
DSX_WAVELET_DB3 = 3
DSX_WAVELET_BANK = 0  # filter bank handed over with dsx_set_wavelet
STAGE_APPROX, STAGE_DETAIL = 0, 1
STREAM_COMPUTE, STREAM_UPLOAD, STREAM_DOWNLOAD = 0, 1, 2
COMM_ID_BYTES = 128
_ERRORS = {-1: "DSX_EINVAL", -2: "DSX_ENOPLAN", -3: "DSX_EHIP", -4: "DSX_ENOMEM", -5: "DSX_ELIMIT",
           -6: "DSX_ECOMM", -7: "DSX_EIO", -8: "DSX_EVALUE"}  # fmt: skip

# every symbol include/dsx.h declares (tests/test_host_native.py checks the list against the header)
EXPORTED_SYMBOLS = [
    "dsx_init", "dsx_destroy", "dsx_last_error", "dsx_device_count", "dsx_plan", "dsx_plan_info", "dsx_set_wavelet", "dsx_graph_stats",
    "dsx_set_shading_device", "dsx_constants_device", "dsx_run_host", "dsx_run_device", "dsx_sync",
    "dsx_malloc", "dsx_free", "dsx_memcpy_h2d", "dsx_memcpy_d2h", "dsx_memcpy_d2d",
    "dsx_timer_start", "dsx_timer_stop", "dsx_profile_enable", "dsx_profile_read",
    "dsx_get_stats", "dsx_get_thresholds", "dsx_get_level", "dsx_set_stop_after", "dsx_set_stack_mode",
    "dsx_bricks_to_planes_u16", "dsx_planes_to_bricks_u16", "dsx_downsample2_u16",
    "dsx_flatfield_correction", "dsx_flatfield_correction_rows", "dsx_foreground_background",
    "dsx_comm_unique_id", "dsx_comm_init", "dsx_comm_destroy", "dsx_comm_broadcast", "dsx_comm_allreduce_f64",
    "dsx_malloc_host", "dsx_free_host", "dsx_memcpy_h2d_async", "dsx_memcpy_d2h_async",
    "dsx_stream_wait", "dsx_stream_sync", "dsx_event_record", "dsx_event_sync",
    "dsx_io_read_chunks", "dsx_io_write_chunks", "dsx_io_write_chunks_blosc", "dsx_blosc_decode", "dsx_blosc_encode",
    "dsx_png_unfilter",
]  # fmt: skip


class DsxError(RuntimeError):
    """Error reported by the HIP engine (code + message of ``dsx_last_error``)."""

    def __init__(self, code, message):
        super().__init__("{} ({}): {}".format(_ERRORS.get(code, "DSX_E?"), code, message))
        self.code = code
        self.message = message


class _Cfg(ctypes.Structure):
    _fields_ = [
        ("wavelet", ctypes.c_int32),
        ("level", ctypes.c_int32),
        ("sigma", ctypes.c_float),
        ("max_threshold", ctypes.c_float),
    ]


class _PlanInfo(ctypes.Structure):
    _fields_ = [
        ("height", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("out_height", ctypes.c_int32),
        ("out_width", ctypes.c_int32),
        ("levels", ctypes.c_int32),
        ("level_h", ctypes.c_int32 * 16),
        ("level_w", ctypes.c_int32 * 16),
        ("fft_len", ctypes.c_int32 * 16),
        ("fft_halo", ctypes.c_int32 * 16),
        ("max_batch", ctypes.c_int32),
        ("workspace_bytes", ctypes.c_uint64),
    ]


_lib = None


def load_library(path=None):
    """Load ``libdsx_hip.so``; raises ``DsxError`` if it has not been built (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise DsxError(
            -3,
            "{} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the destripe engine has no CPU fallback)".format(p),
        )
    lib = ctypes.CDLL(p)
    vp, i32, f32p = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)
    lib.dsx_init.argtypes = [i32, ctypes.POINTER(vp)]
    lib.dsx_destroy.argtypes = [vp]
    lib.dsx_destroy.restype = None
    lib.dsx_last_error.argtypes = [vp]
    lib.dsx_last_error.restype = ctypes.c_char_p
    lib.dsx_device_count.argtypes = []
    lib.dsx_plan.argtypes = [vp, i32, i32, i32, ctypes.POINTER(_Cfg), ctypes.POINTER(_Cfg),
                             ctypes.c_double, vp, vp, i32, i32]  # fmt: skip
    lib.dsx_plan_info.argtypes = [vp, ctypes.POINTER(_PlanInfo)]
    lib.dsx_set_shading_device.argtypes = [vp, vp, vp, i32, i32]
    lib.dsx_constants_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    lib.dsx_run_host.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    lib.dsx_run_device.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    lib.dsx_sync.argtypes = [vp]
    lib.dsx_malloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.dsx_free.argtypes = [vp, vp]
    lib.dsx_memcpy_h2d.argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.dsx_memcpy_d2h.argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.dsx_memcpy_d2d.argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.dsx_timer_start.argtypes = [vp]
    lib.dsx_timer_stop.argtypes = [vp, f32p]
    lib.dsx_profile_enable.argtypes = [vp, i32]
    lib.dsx_profile_read.argtypes = [vp, i32, f32p, ctypes.POINTER(ctypes.c_int32),
                                     ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i32)]  # fmt: skip
    lib.dsx_get_stats.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double),
                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]  # fmt: skip
    lib.dsx_get_thresholds.argtypes = [vp, i32, i32, f32p, f32p]
    lib.dsx_get_level.argtypes = [vp, i32, i32, i32, vp]
    lib.dsx_set_stop_after.argtypes = [vp, i32]
    lib.dsx_set_stack_mode.argtypes = [vp, i32]
    lib.dsx_graph_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.dsx_set_wavelet.argtypes = [vp] + [ctypes.POINTER(ctypes.c_double)] * 4 + [i32]
    lib.dsx_bricks_to_planes_u16.argtypes = [vp, vp, vp] + [i32] * 7
    lib.dsx_planes_to_bricks_u16.argtypes = [vp, vp, vp] + [i32] * 7
    lib.dsx_downsample2_u16.argtypes = [vp, vp, vp, i32, i32, i32]
    lib.dsx_foreground_background.argtypes = [vp, vp, i32, ctypes.c_size_t, ctypes.c_float,
                                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), vp]  # fmt: skip
    lib.dsx_flatfield_correction.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, i32, ctypes.c_float, vp]
    lib.dsx_flatfield_correction_rows.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, i32, ctypes.c_float, vp, vp]
    lib.dsx_comm_unique_id.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    lib.dsx_comm_init.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, i32, i32]
    lib.dsx_comm_destroy.argtypes = [vp]
    lib.dsx_comm_broadcast.argtypes = [vp, vp, ctypes.c_size_t, i32]
    lib.dsx_comm_allreduce_f64.argtypes = [vp, ctypes.POINTER(ctypes.c_double), i32, i32]
    lib.dsx_malloc_host.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.dsx_free_host.argtypes = [vp, vp]
    lib.dsx_memcpy_h2d_async.argtypes = [vp, vp, vp, ctypes.c_size_t, i32]
    lib.dsx_memcpy_d2h_async.argtypes = [vp, vp, vp, ctypes.c_size_t, i32]
    lib.dsx_stream_wait.argtypes = [vp, i32, i32]
    lib.dsx_stream_sync.argtypes = [vp, i32]
    lib.dsx_event_record.argtypes = [vp, i32, i32]
    lib.dsx_event_sync.argtypes = [vp, i32]
    lib.dsx_io_read_chunks.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                       ctypes.POINTER(ctypes.c_size_t), i32, i32, i32, ctypes.c_uint16]  # fmt: skip
    lib.dsx_io_write_chunks.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                        ctypes.POINTER(ctypes.c_size_t), i32, i32, i32]  # fmt: skip
    lib.dsx_io_write_chunks_blosc.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                              ctypes.POINTER(ctypes.c_size_t), i32, i32, i32, i32, i32]  # fmt: skip
    lib.dsx_png_unfilter.argtypes = [vp, i32, i32, i32]
    lib.dsx_blosc_decode.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t]
    lib.dsx_blosc_encode.argtypes = [vp, ctypes.c_size_t, i32, i32, i32, vp, ctypes.c_size_t,
                                     ctypes.POINTER(ctypes.c_size_t)]  # fmt: skip
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("dsx_destroy", "dsx_last_error"):
            fn.restype = ctypes.c_int
    if path is None:
        _lib = lib
    return lib


def _wavelet_key(cfg):
    w = cfg.get("wavelet", "db3")
    return w.lower() if isinstance(w, str) else w


def _as_cfg(cfg):
    """Reference config dict {"wavelet","level","sigma","max_threshold"} -> C struct."""
    wid = DSX_WAVELET_DB3 if _wavelet_key(cfg) == "db3" and not os.environ.get("DSX_GENERIC_DB3") else DSX_WAVELET_BANK
    level = cfg.get("level", 0)
    return _Cfg(wid, -1 if level is None else int(level), float(cfg.get("sigma", 64)),
                float(cfg.get("max_threshold", 4)))  # fmt: skip


class DeviceBuffer:
    """A device allocation owned by an engine."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        engine._check(engine._lib.dsx_malloc(engine._ctx, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr is not None and self.engine._ctx is not None:
            self.engine._lib.dsx_free(self.engine._ctx, ctypes.c_void_p(self.ptr))
        self.ptr = None

    def upload(self, array, offset=0):
        a = np.ascontiguousarray(array)
        assert offset + a.nbytes <= self.nbytes
        self.engine._check(
            self.engine._lib.dsx_memcpy_h2d(self.engine._ctx, ctypes.c_void_p(self.ptr + offset),
                                            a.ctypes.data_as(ctypes.c_void_p), a.nbytes)
        )  # fmt: skip

    def download(self, shape, dtype, offset=0):
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        self.engine._check(
            self.engine._lib.dsx_memcpy_d2h(self.engine._ctx, out.ctypes.data_as(ctypes.c_void_p),
                                            ctypes.c_void_p(self.ptr + offset), out.nbytes)
        )  # fmt: skip
        return out


class PinnedBuffer:
    """Page-locked host memory owned by an engine (``dsx_malloc_host``): async copies need it."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        engine._check(engine._lib.dsx_malloc_host(engine._ctx, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def array(self, shape, dtype, offset=0):
        """NumPy view of (part of) the buffer."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert offset + n <= self.nbytes
        raw = (ctypes.c_char * n).from_address(self.ptr + offset)
        return np.frombuffer(raw, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr is not None and self.engine._ctx is not None:
            self.engine._lib.dsx_free_host(self.engine._ctx, ctypes.c_void_p(self.ptr))
        self.ptr = None


def _dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.uint16:
        return DSX_U16
    if dtype == np.float32:
        return DSX_F32
    raise ValueError("planes must be uint16 or float32, got {}".format(dtype))


class DestripeEngine:
    """One context on one GPU: ``plan()`` once per plane geometry / config pair, then ``run()``."""

    def __init__(self, device=0):
        self._lib = load_library()
        ctx = ctypes.c_void_p()
        rc = self._lib.dsx_init(int(device), ctypes.byref(ctx))
        if rc != 0:
            msg = self._lib.dsx_last_error(None)
            raise DsxError(rc, msg.decode() if msg else "dsx_init failed")
        self._ctx = ctx
        self.device = int(device)
        self.info = None

    # -- plumbing --------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.dsx_last_error(self._ctx)
            if rc == -8:  # DSX_EVALUE: the reference's own exception type and message (numpy.histogram)
                raise ValueError(msg.decode() if msg else "autodetected range of [nan, nan] is not finite")
            raise DsxError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_ctx", None) is not None:
            self._lib.dsx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plan ------------------------------------------------------------------------------------
    def plan(self, height, width, cells_config, no_cells_config, microscope_high_int=2700,
             max_batch=32, flatfield=None, darkfield=None):  # fmt: skip
        cells, no_cells = _as_cfg(cells_config), _as_cfg(no_cells_config)
        # Both configs go through ONE decomposition: which config a plane takes is only known once the
        # statistic fused into the level-1 analysis is (filtering.py:455-462 decides before it transforms).
        if _wavelet_key(cells_config) != _wavelet_key(no_cells_config):
            raise ValueError("cells_config and no_cells_config must name the same wavelet")
        if cells.wavelet == DSX_WAVELET_BANK:  # anything but db3: hand the filter bank over (DSX_GENERIC_DB3: db3 too)
            bank = [np.ascontiguousarray(f, dtype=np.float64) for f in wavelets.filter_bank(_wavelet_key(cells_config))]
            dp = ctypes.POINTER(ctypes.c_double)
            rc = self._lib.dsx_set_wavelet(self._ctx, *[f.ctypes.data_as(dp) for f in bank], len(bank[0]))
            if rc == -1:
                raise ValueError(self._lib.dsx_last_error(self._ctx).decode())
            self._check(rc)

        def call(flat_p, dark_p, dark_h, dark_w):
            rc = self._lib.dsx_plan(self._ctx, int(height), int(width), int(max_batch), ctypes.byref(cells),
                                    ctypes.byref(no_cells), float(microscope_high_int), flat_p, dark_p,
                                    int(dark_h), int(dark_w))  # fmt: skip
            if rc == -1:  # the reference raises ValueError for these (filtering.py:107-112, 379-391)
                raise ValueError(self._lib.dsx_last_error(self._ctx).decode())
            self._check(rc)
            info = _PlanInfo()
            self._check(self._lib.dsx_plan_info(self._ctx, ctypes.byref(info)))
            return info

        if (flatfield is None) != (darkfield is None):
            raise ValueError("flatfield and darkfield must be given together")
        info = call(None, None, 0, 0)
        if flatfield is not None:
            flat = np.ascontiguousarray(flatfield, dtype=np.float32)
            dark = np.ascontiguousarray(darkfield, dtype=np.float32)
            out_shape = (info.out_height, info.out_width)
            if flat.ndim != 2 or dark.ndim != 2:
                raise ValueError("flatfield / darkfield must be 2-D planes")
            # same checks (and messages) as flatfield_correction(), filtering.py:377-391
            cropped = dark[: out_shape[0], : out_shape[1]].shape
            if cropped != out_shape:
                raise ValueError(
                    "Please, check the shape of the darkfield. "
                    "Image: {} - Darkfield: {}".format(out_shape, cropped)
                )
            if flat.shape != out_shape:
                raise ValueError(
                    "Please, check the shape of the flatfield."
                    "Image: {} - Flatfield: {}".format(out_shape, flat.shape)
                )
            info = call(flat.ctypes.data_as(ctypes.c_void_p), dark.ctypes.data_as(ctypes.c_void_p),
                        dark.shape[0], dark.shape[1])  # fmt: skip
        self.info = info
        return info

    def graph_stats(self):
        """``(graph launches, captures)`` of this context (``dsx_graph_stats``)."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.dsx_graph_stats(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    @property
    def out_shape(self):
        return (self.info.out_height, self.info.out_width)

    @property
    def levels(self):
        return self.info.levels

    def level_shape(self, level):
        return (self.info.level_h[level], self.info.level_w[level])

    # -- run -------------------------------------------------------------------------------------
    def run(self, planes, out_dtype=np.float32, return_cfg=False):
        """Host arrays in, host arrays out: ``planes[n, H, W]`` (uint16 / float32)."""
        a = np.ascontiguousarray(planes)
        if a.ndim == 2:
            a = a[None]
        if a.ndim != 3 or a.shape[1:] != (self.info.height, self.info.width):
            raise ValueError("planes must be [n, {}, {}]".format(self.info.height, self.info.width))
        n = a.shape[0]
        out = np.empty((n,) + self.out_shape, dtype=out_dtype)
        cfg = np.zeros(n, dtype=np.int32)
        self._check(
            self._lib.dsx_run_host(self._ctx, a.ctypes.data_as(ctypes.c_void_p), _dtype_code(a.dtype), n,
                                   out.ctypes.data_as(ctypes.c_void_p), _dtype_code(out.dtype),
                                   cfg.ctypes.data_as(ctypes.c_void_p))
        )  # fmt: skip
        return (out, cfg) if return_cfg else out

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def run_device(self, d_in, in_dtype, n, d_out, out_dtype, d_cfg=None):
        """Device buffers; asynchronous on the engine stream."""
        self._check(
            self._lib.dsx_run_device(self._ctx, ctypes.c_void_p(d_in.ptr), _dtype_code(in_dtype), int(n),
                                     ctypes.c_void_p(d_out.ptr), _dtype_code(out_dtype),
                                     ctypes.c_void_p(d_cfg.ptr) if d_cfg is not None else None)
        )  # fmt: skip

    def sync(self):
        self._check(self._lib.dsx_sync(self._ctx))

    def timer_start(self):
        self._check(self._lib.dsx_timer_start(self._ctx))

    def timer_stop(self):
        ms = ctypes.c_float()
        self._check(self._lib.dsx_timer_stop(self._ctx, ctypes.byref(ms)))
        return ms.value

    def constants_device(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._lib.dsx_constants_device(self._ctx, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    # -- multi-GPU: RCCL communicator (include/dsx.h, dsx_comm_*) --------------------------------
    def comm_unique_id(self):
        """128-byte RCCL unique id; rank 0 creates it and hands it to the other ranks (any host channel)."""
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        self._check(self._lib.dsx_comm_unique_id(self._ctx, buf, COMM_ID_BYTES))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """Collective over all ranks of the job."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("the RCCL unique id has {} bytes".format(COMM_ID_BYTES))
        self._check(self._lib.dsx_comm_init(self._ctx, bytes(unique_id), COMM_ID_BYTES, int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._lib.dsx_comm_destroy(self._ctx))

    def comm_broadcast(self, d_ptr, nbytes, root=0):
        """In-place RCCL broadcast of device memory (address or DeviceBuffer) from ``root``."""
        ptr = d_ptr.ptr if isinstance(d_ptr, DeviceBuffer) else d_ptr
        self._check(self._lib.dsx_comm_broadcast(self._ctx, ctypes.c_void_p(ptr), int(nbytes), int(root)))

    def comm_allreduce(self, values, op="sum"):
        """All-reduce of a few host doubles over the ranks (also a barrier): op sum / max / min."""
        v = (ctypes.c_double * len(values))(*[float(x) for x in values])
        self._check(self._lib.dsx_comm_allreduce_f64(self._ctx, v, len(values), {"sum": 0, "max": 1, "min": 2}[op]))
        return [float(x) for x in v]

    # -- pinned staging + copy streams (overlapped chunk map) ------------------------------------
    def alloc_host(self, nbytes):
        return PinnedBuffer(self, nbytes)

    def copy_h2d_async(self, d_buf, host_array, stream=STREAM_UPLOAD, offset=0):
        a = host_array
        assert a.flags["C_CONTIGUOUS"] and offset + a.nbytes <= d_buf.nbytes
        self._check(self._lib.dsx_memcpy_h2d_async(self._ctx, ctypes.c_void_p(d_buf.ptr + offset),
                                                   a.ctypes.data_as(ctypes.c_void_p), a.nbytes, int(stream)))  # fmt: skip

    def copy_d2h_async(self, host_array, d_buf, stream=STREAM_DOWNLOAD, offset=0):
        a = host_array
        assert a.flags["C_CONTIGUOUS"] and offset + a.nbytes <= d_buf.nbytes
        self._check(self._lib.dsx_memcpy_d2h_async(self._ctx, a.ctypes.data_as(ctypes.c_void_p),
                                                   ctypes.c_void_p(d_buf.ptr + offset), a.nbytes, int(stream)))  # fmt: skip

    def stream_wait(self, waiter, signaller):
        self._check(self._lib.dsx_stream_wait(self._ctx, int(waiter), int(signaller)))

    def stream_sync(self, stream):
        self._check(self._lib.dsx_stream_sync(self._ctx, int(stream)))

    # -- chunk files on native threads (dsx_io.h); ctypes releases the GIL for the call -----------
    def io_read_chunks(self, paths, arrays, threads=16, zlib_chunks=False, fill_value=0, codec=None):
        """Chunk files ``paths[i]`` -> ``arrays[i]`` (C-contiguous NumPy arrays of the decompressed chunk size).
        ``codec``: ``DSX_CODEC_*`` (0 raw, 1 zlib, 2 Blosc); default from ``zlib_chunks``."""
        n = len(paths)
        cp = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        dp = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrays])
        nb = (ctypes.c_size_t * n)(*[a.nbytes for a in arrays])
        code = (1 if zlib_chunks else 0) if codec is None else int(codec)
        self._check(self._lib.dsx_io_read_chunks(self._ctx, cp, dp, nb, n, int(threads), code, int(fill_value)))

    def io_write_chunks(self, paths, arrays, threads=16, zlib_level=-1, blosc=None):
        """``arrays[i]`` -> chunk files ``paths[i]`` (raw, zlib streams for ``zlib_level >= 0``, or Blosc-zstd frames for
        ``blosc = (clevel, typesize, shuffle)``), atomically."""
        n = len(paths)
        cp = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        dp = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrays])
        nb = (ctypes.c_size_t * n)(*[a.nbytes for a in arrays])
        if blosc is not None:
            clevel, typesize, shuffle = blosc
            self._check(self._lib.dsx_io_write_chunks_blosc(self._ctx, cp, dp, nb, n, int(threads), int(clevel),
                                                            int(typesize), 1 if shuffle else 0))  # fmt: skip
            return
        self._check(self._lib.dsx_io_write_chunks(self._ctx, cp, dp, nb, n, int(threads), int(zlib_level)))

    def event_record(self, slot, stream):
        self._check(self._lib.dsx_event_record(self._ctx, int(slot), int(stream)))

    def event_sync(self, slot):
        self._check(self._lib.dsx_event_sync(self._ctx, int(slot)))

    def copy_d2d_async(self, d_dst, d_src, nbytes):
        self._check(self._lib.dsx_memcpy_d2d(self._ctx, ctypes.c_void_p(d_dst.ptr), ctypes.c_void_p(d_src.ptr), int(nbytes)))

    def profile(self, on):
        self._check(self._lib.dsx_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self):
        ms = (ctypes.c_float * 16)()
        cnt = (ctypes.c_int32 * 16)()
        names = (ctypes.c_char_p * 16)()
        n = ctypes.c_int()
        self._check(self._lib.dsx_profile_read(self._ctx, 16, ms, cnt, names, ctypes.byref(n)))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n.value)}

    # -- data formats either side of the filter (device buffers, asynchronous) -------------------
    def bricks_to_planes(self, d_bricks, d_planes, zyx, brick, z0=0):
        """Zarr chunk order -> dense ``[Z, H, W]`` uint16 (``zarr_destriper.py:1066-1074``)."""
        self._check(self._lib.dsx_bricks_to_planes_u16(self._ctx, ctypes.c_void_p(d_bricks.ptr),
                                                       ctypes.c_void_p(d_planes.ptr), *map(int, zyx),
                                                       *map(int, brick), int(z0)))  # fmt: skip

    def planes_to_bricks(self, d_planes, d_bricks, zyx, brick, z0=0):
        """Dense ``[Z, H, W]`` uint16 -> Zarr chunk order, 0 outside the stack (``zarr_destriper.py:336``)."""
        self._check(self._lib.dsx_planes_to_bricks_u16(self._ctx, ctypes.c_void_p(d_planes.ptr),
                                                       ctypes.c_void_p(d_bricks.ptr), *map(int, zyx),
                                                       *map(int, brick), int(z0)))  # fmt: skip

    def downsample2(self, d_src, d_dst, zyx):
        """One 2x2x2 windowed-mean pyramid level, uint16 (``zarr_destriper.py:365-407``)."""
        rc = self._lib.dsx_downsample2_u16(self._ctx, ctypes.c_void_p(d_src.ptr), ctypes.c_void_p(d_dst.ptr),
                                           *map(int, zyx))  # fmt: skip
        if rc == -1:
            raise ValueError(self._lib.dsx_last_error(self._ctx).decode())
        self._check(rc)

    def foreground_background(self, image, cutoff, want_mask=True):
        """``(fore_mean, back_mean, mask uint8)`` of a host image (uint16 / float32, any shape)."""
        a = np.ascontiguousarray(image)
        d_img = self.alloc(max(a.nbytes, 16))
        d_mask = self.alloc(max(a.size, 16)) if want_mask else None
        try:
            d_img.upload(a)
            f, b = ctypes.c_double(), ctypes.c_double()
            rc = self._lib.dsx_foreground_background(self._ctx, ctypes.c_void_p(d_img.ptr), _dtype_code(a.dtype), a.size,
                                                     float(cutoff), ctypes.byref(f), ctypes.byref(b),
                                                     ctypes.c_void_p(d_mask.ptr) if want_mask else None)  # fmt: skip
            if rc == -1:
                raise ValueError(self._lib.dsx_last_error(self._ctx).decode())
            self._check(rc)
            mask = d_mask.download(a.shape, np.uint8) if want_mask else None
            return f.value, b.value, mask
        finally:
            d_img.free()
            if d_mask is not None:
                d_mask.free()

    def flatfield_correction(self, plane, flatfield, darkfield, baseline=0.0):
        """One host plane (uint16 / float32) through ``dsx_flatfield_correction[_rows]``; uint16 result.
        ``baseline``: a scalar, or one value per plane row."""
        a = np.ascontiguousarray(plane)
        flat = np.ascontiguousarray(flatfield, dtype=np.float32)
        dark = np.ascontiguousarray(darkfield, dtype=np.float32)
        H, W = a.shape
        rows = None
        if np.ndim(baseline) > 0:
            rows = np.ascontiguousarray(baseline, dtype=np.float32).ravel()
            if rows.size != H:
                raise ValueError("a per-row baseline needs one value per plane row ({} != {})".format(rows.size, H))
            baseline = 0.0
        bufs = [self.alloc(max(x.nbytes, 16)) for x in (a, flat, dark)] + [self.alloc(max(H * W * 2, 16))]
        if rows is not None:
            bufs.append(self.alloc(max(rows.nbytes, 16)))
        try:
            for b, x in zip(bufs, (a, flat, dark)):
                b.upload(x)
            if rows is not None:
                bufs[4].upload(rows)
            rc = self._lib.dsx_flatfield_correction_rows(self._ctx, ctypes.c_void_p(bufs[0].ptr), _dtype_code(a.dtype), H, W,
                                                         ctypes.c_void_p(bufs[1].ptr), ctypes.c_void_p(bufs[2].ptr),
                                                         dark.shape[0], dark.shape[1], float(baseline),
                                                         ctypes.c_void_p(bufs[4].ptr) if rows is not None else None,
                                                         ctypes.c_void_p(bufs[3].ptr))  # fmt: skip
            if rc == -1:
                raise ValueError(self._lib.dsx_last_error(self._ctx).decode())
            self._check(rc)
            return bufs[3].download((H, W), np.uint16)
        finally:
            for b in bufs:
                b.free()

    # -- parity hooks ----------------------------------------------------------------------------
    def set_stack_mode(self, on):
        """Planes of one ``run`` call share one Otsu threshold per level (the reference's 3-D input mode)."""
        self._check(self._lib.dsx_set_stack_mode(self._ctx, 1 if on else 0))

    def set_stop_after(self, stage):
        self._check(self._lib.dsx_set_stop_after(self._ctx, int(stage)))

    def stats(self, plane):
        f, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
        self._check(self._lib.dsx_get_stats(self._ctx, int(plane), ctypes.byref(f), ctypes.byref(b), ctypes.byref(c)))
        return f.value, b.value, c.value

    def thresholds(self, plane, level):
        o, t = ctypes.c_float(), ctypes.c_float()
        self._check(self._lib.dsx_get_thresholds(self._ctx, int(plane), int(level), ctypes.byref(o), ctypes.byref(t)))
        return o.value, t.value

    def level_array(self, plane, level, stage=STAGE_DETAIL):
        out = np.empty(self.level_shape(level), dtype=np.float32)
        self._check(self._lib.dsx_get_level(self._ctx, int(plane), int(level), int(stage),
                                            out.ctypes.data_as(ctypes.c_void_p)))  # fmt: skip
        return out
