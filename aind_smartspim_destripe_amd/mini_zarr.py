"""Minimal Zarr-v2 directory store (read / write) for the chunk map.

Neither ``zarr`` nor ``numcodecs`` is installed in this environment, so the chunk map carries its own small
implementation of the subset it needs: C-order arrays, little-endian numeric dtypes, ``compressor`` ``null``,
``zlib`` or ``blosc``, ``"."`` or ``"/"`` chunk-key separators, basic slicing with unit steps.  The reference's
production arrays are ``uint16``, chunks ``(1, 1, 64, 128, 128)``, ``Blosc(cname="zstd", clevel=3, shuffle=SHUFFLE)``,
``dimension_separator="/"`` (``zarr_destriper.py:1066-1074``).  Blosc frames are decoded / encoded by the native
library (``csrc/dsx_io.h``: the c-blosc 1.x container restated from its format description, ``libzstd.so.1`` /
``liblz4.so.1`` of the image ``dlopen``ed; frames with zstd, lz4, blosclz or zlib inside, byte- or bit-shuffled, are
read -- Zarr's own default ``Blosc(lz4, 5, SHUFFLE)`` included -- and zstd frames are written).  Pinned by frames of the
real c-blosc 1.21.0 (``tests/golden/blosc_frames.npz``, ``oracle/make_golden_blosc.py``).
"""

import itertools
import json
import os
import zlib

import numpy as np

CODEC_RAW, CODEC_ZLIB, CODEC_BLOSC = 0, 1, 2  # DSX_CODEC_* of include/dsx.h
# numcodecs.Blosc(cname="zstd", clevel=3, shuffle=Blosc.SHUFFLE).get_config(), zarr_destriper.py:1066-1074
BLOSC_ZSTD = {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 1, "blocksize": 0}


def _native():
    from . import engine

    return engine.load_library()


def blosc_decode(frame, nbytes):
    """One Blosc frame -> ``bytes`` object of ``nbytes`` (``dsx_blosc_decode``)."""
    import ctypes

    lib = _native()
    out = ctypes.create_string_buffer(nbytes)
    rc = lib.dsx_blosc_decode(frame, len(frame), out, nbytes)
    if rc != 0:
        raise ValueError((lib.dsx_last_error(None) or b"blosc: decode failed").decode())
    return out.raw


def blosc_encode(raw, typesize, clevel=3, shuffle=True):
    """Bytes-like -> one Blosc frame with zstd inside (``dsx_blosc_encode``)."""
    import ctypes

    lib = _native()
    raw = bytes(raw)
    out = ctypes.create_string_buffer(len(raw) + 16)
    n = ctypes.c_size_t()
    rc = lib.dsx_blosc_encode(raw, len(raw), int(typesize), int(clevel), 1 if shuffle else 0, out, len(raw) + 16,
                              ctypes.byref(n))  # fmt: skip
    if rc != 0:
        raise ValueError((lib.dsx_last_error(None) or b"blosc: encode failed").decode())
    return out.raw[: n.value]


class MiniZarrArray:
    def __init__(self, path, meta):
        self.path = str(path)
        self.shape = tuple(meta["shape"])
        self.chunks = tuple(meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.fill_value = meta.get("fill_value", 0) or 0
        self.sep = meta.get("dimension_separator", ".")
        comp = meta.get("compressor")
        self.compressor_meta = None if comp is None else dict(comp)  # as written in .zarray (matches())
        if comp is None:
            self.compressor = None
        elif comp.get("id") == "zlib":
            self.compressor = ("zlib", int(comp.get("level", 1)))
        elif comp.get("id") == "blosc":
            # reading only needs the frames (their headers name the inner codec); writing needs cname zstd
            self.compressor = ("blosc", int(comp.get("clevel", 5)), str(comp.get("cname", "lz4")),
                               int(comp.get("shuffle", 1)))  # fmt: skip
        else:
            raise NotImplementedError(
                "compressor {!r} is not available (null / zlib / blosc are)".format(comp.get("id"))
            )
        if meta.get("order", "C") != "C" or meta.get("filters"):
            raise NotImplementedError("only C-order arrays without filters are supported")
        self.ndim = len(self.shape)
        self._dirs_made = set()

    @property
    def codec(self):
        """``DSX_CODEC_*`` of the chunk files (``dsx_io_read_chunks``)."""
        return {None: CODEC_RAW, "zlib": CODEC_ZLIB, "blosc": CODEC_BLOSC}[self.compressor and self.compressor[0]]

    def blosc_write_params(self):
        """``(clevel, typesize, byte shuffle)`` for ``dsx_io_write_chunks_blosc``; raises for what the writer lacks."""
        _, clevel, cname, shuffle = self.compressor
        if cname != "zstd":
            raise NotImplementedError("Blosc frames are written with zstd inside; cname {!r} is read-only".format(cname))
        if shuffle == 2 or (shuffle == -1 and self.dtype.itemsize == 1):
            raise NotImplementedError("Blosc bit-shuffle is not implemented")
        return clevel, self.dtype.itemsize, shuffle != 0

    def _decode(self, raw):
        if self.compressor is None:
            return raw
        if self.compressor[0] == "zlib":
            return zlib.decompress(raw)
        return blosc_decode(raw, int(np.prod(self.chunks)) * self.dtype.itemsize)

    def _encode(self, raw):
        if self.compressor is None:
            return raw
        if self.compressor[0] == "zlib":
            return zlib.compress(raw, self.compressor[1])
        clevel, typesize, shuffle = self.blosc_write_params()
        return blosc_encode(raw, typesize, clevel, shuffle)

    # -- construction -------------------------------------------------------------------------
    @staticmethod
    def _compressor_meta(compressor):
        """The ``compressor`` entry of ``.zarray`` for what :meth:`create` accepts."""
        if compressor is None:
            return None
        if compressor == "zlib":
            return {"id": "zlib", "level": 1}
        if compressor == "blosc":
            return dict(BLOSC_ZSTD)
        if isinstance(compressor, dict):
            return dict(compressor)  # a numcodecs get_config() dict
        raise NotImplementedError("compressors: None, 'zlib', 'blosc' or a numcodecs config dict")

    @classmethod
    def create(cls, path, shape, chunks, dtype, compressor=None, dimension_separator="/", fill_value=0,
               overwrite=True):  # fmt: skip
        os.makedirs(path, exist_ok=True)
        if not overwrite and os.path.exists(os.path.join(path, ".zarray")):
            raise FileExistsError(path)
        comp = cls._compressor_meta(compressor)
        meta = {
            "zarr_format": 2,
            "shape": list(shape),
            "chunks": list(chunks),
            "dtype": np.dtype(dtype).str,
            "compressor": comp,
            "fill_value": fill_value,
            "order": "C",
            "filters": None,
            "dimension_separator": dimension_separator,
        }
        # written to a temporary name and renamed: a concurrent reader (another rank polling for the array)
        # sees the metadata either whole or not at all
        tmp = os.path.join(path, ".zarray.tmp.{}".format(os.getpid()))
        with open(tmp, "w") as f:
            json.dump(meta, f)
        os.replace(tmp, os.path.join(path, ".zarray"))
        return cls(path, meta)

    @classmethod
    def open(cls, path):
        with open(os.path.join(path, ".zarray")) as f:
            return cls(path, json.load(f))

    def matches(self, shape, chunks, dtype, compressor=Ellipsis):
        """Does this array have the given geometry -- and, when ``compressor`` is given (what :meth:`create` takes),
        the given codec?  Tells the array rank 0 has just created from a stale one: a left-over of an earlier run with
        ANOTHER codec has the same geometry, and a rank that opened it would write chunks the new metadata cannot read."""
        ok = (self.shape == tuple(int(x) for x in shape) and self.chunks == tuple(int(x) for x in chunks)
              and self.dtype == np.dtype(dtype))
        if ok and compressor is not Ellipsis:
            ok = self.compressor_meta == self._compressor_meta(compressor)
        return ok

    # -- chunk io -----------------------------------------------------------------------------
    def _chunk_path(self, idx):
        return os.path.join(self.path, self.sep.join(str(i) for i in idx).replace("/", os.sep))

    def _read_chunk(self, idx):
        p = self._chunk_path(idx)
        if not os.path.exists(p):
            return np.full(self.chunks, self.fill_value, dtype=self.dtype)
        with open(p, "rb") as f:
            raw = self._decode(f.read())
        return np.frombuffer(raw, dtype=self.dtype).reshape(self.chunks).copy()

    def read_chunk_into(self, idx, out_flat):
        """Decompressed chunk ``idx`` (the whole brick, as stored) into a flat array of chunk size."""
        p = self._chunk_path(idx)
        try:
            f = open(p, "rb", buffering=0)
        except FileNotFoundError:
            out_flat[...] = self.fill_value
            return
        with f:
            if self.compressor is None:
                # raw chunk: straight into the (pinned) staging buffer, one copy, no GIL held by the read
                view = memoryview(out_flat).cast("B")
                got = f.readinto(view)
                if got != len(view):
                    raise ValueError("chunk {} has {} bytes, expected {}".format(p, got, len(view)))
                return
            raw = self._decode(f.read())
        out_flat[...] = np.frombuffer(raw, dtype=self.dtype)

    def write_chunk_flat(self, idx, flat):
        """Store a whole brick given in chunk (C) order."""
        self._write_chunk(idx, np.asarray(flat, dtype=self.dtype))

    def _write_chunk(self, idx, block):
        p = self._chunk_path(idx)
        d = os.path.dirname(p)
        if d not in self._dirs_made:
            os.makedirs(d, exist_ok=True)
            self._dirs_made.add(d)
        raw = memoryview(np.ascontiguousarray(block, dtype=self.dtype)).cast("B")
        raw = self._encode(raw)
        with open(p + ".tmp", "wb", buffering=0) as f:
            f.write(raw)
        os.replace(p + ".tmp", p)

    # -- slicing ------------------------------------------------------------------------------
    def _normalize(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (self.ndim - len(key) + 1) + key[i + 1 :]
        key = key + (slice(None),) * (self.ndim - len(key))
        out, squeeze = [], []
        for ax, (k, n) in enumerate(zip(key, self.shape)):
            if isinstance(k, (int, np.integer)):
                k = int(k) + (n if k < 0 else 0)
                out.append((k, k + 1))
                squeeze.append(ax)
            else:
                start, stop, step = k.indices(n)
                if step != 1:
                    raise NotImplementedError("only unit steps")
                out.append((start, max(start, stop)))
        return out, tuple(squeeze)

    def _chunk_ranges(self, bounds):
        return [range(lo // c, (hi - 1) // c + 1) if hi > lo else range(0) for (lo, hi), c in zip(bounds, self.chunks)]

    def __getitem__(self, key):
        bounds, squeeze = self._normalize(key)
        out = np.empty([hi - lo for lo, hi in bounds], dtype=self.dtype)
        for idx in itertools.product(*self._chunk_ranges(bounds)):
            chunk = self._read_chunk(idx)
            src, dst = [], []
            for (lo, hi), c, i in zip(bounds, self.chunks, idx):
                a, b = max(lo, i * c), min(hi, (i + 1) * c)
                src.append(slice(a - i * c, b - i * c))
                dst.append(slice(a - lo, b - lo))
            out[tuple(dst)] = chunk[tuple(src)]
        return out.squeeze(axis=squeeze) if squeeze else out

    def __setitem__(self, key, value):
        bounds, _ = self._normalize(key)
        shape = [hi - lo for lo, hi in bounds]
        # NumPy assignment cast, as zarr does: float -> uint16 truncates (zarr_destriper.py:336)
        value = np.broadcast_to(np.asarray(value), shape) if np.ndim(value) else np.full(shape, value)
        value = value.astype(self.dtype, copy=False)
        for idx in itertools.product(*self._chunk_ranges(bounds)):
            src, dst, full = [], [], True
            for (lo, hi), c, i, n in zip(bounds, self.chunks, idx, self.shape):
                a, b = max(lo, i * c), min(hi, (i + 1) * c)
                dst.append(slice(a - i * c, b - i * c))
                src.append(slice(a - lo, b - lo))
                full = full and (a == i * c) and (b == min((i + 1) * c, n))
            if full and all(s.stop - s.start == c for s, c in zip(dst, self.chunks)):
                chunk = value[tuple(src)]
            else:
                chunk = self._read_chunk(idx)
                chunk[tuple(dst)] = value[tuple(src)]
            self._write_chunk(idx, chunk)
