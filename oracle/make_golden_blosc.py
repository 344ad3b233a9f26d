"""Writes tests/golden/blosc_frames.npz with the real c-blosc (TEST INFRASTRUCTURE ONLY).

Run in this container (either interpreter; the library is loaded with ctypes):

    python oracle/make_golden_blosc.py

``/opt/conda/lib/libblosc.so.1`` is c-blosc 1.21.0 -- the library numcodecs.Blosc wraps, i.e. what the reference reads and
writes its Zarr chunks with (``zarr_destriper.py:1066-1074``: ``Blosc(cname="zstd", clevel=3, shuffle=Blosc.SHUFFLE)``).
numcodecs itself is not installed, so the frames are made by calling ``blosc_compress_ctx`` directly: every inner codec
the build has (blosclz, lz4, lz4hc, zlib, zstd), no / byte / bit shuffle, type sizes 1 ... 8, automatic and forced block
sizes (many blocks, a short last block, split and unsplit streams), buffers below the 128-byte minimum, incompressible
data (stored frames).  The frames pin the native reader (``csrc/dsx_io.h: blosc_decode``); the payloads are regenerated
from the stored seeds by ``tests/test_blosc.py::payload``.
"""

import ctypes
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "blosc_frames.npz")
LIB = "/opt/conda/lib/libblosc.so.1"


def payload(kind, seed, nbytes):
    """Test payloads (the same function lives in tests/test_blosc.py)."""
    rs = np.random.RandomState(seed)
    if kind == "brick":  # smooth uint16 ramp + noise, like a chunk of image planes
        n = nbytes // 2
        v = (np.arange(n) % 977) * 13 + rs.randint(0, 40, n) + 300
        b = v.astype("<u2").tobytes()
    elif kind == "noise":  # incompressible
        b = rs.bytes(nbytes)
    elif kind == "runs":  # long runs and short repeats
        b = np.repeat(rs.randint(0, 256, nbytes // 37 + 1).astype(np.uint8), 37).tobytes()
    elif kind == "f32":
        b = np.cumsum(rs.standard_normal(nbytes // 4 + 1).astype(np.float32)).astype("<f4").tobytes()
    else:
        raise ValueError(kind)
    return (b + bytes(nbytes))[:nbytes]


def main():
    lib = ctypes.CDLL(LIB)
    lib.blosc_get_version_string.restype = ctypes.c_char_p
    lib.blosc_compress_ctx.restype = ctypes.c_int
    lib.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    lib.blosc_decompress_ctx.restype = ctypes.c_int
    lib.blosc_decompress_ctx.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    version = lib.blosc_get_version_string().decode()

    cases = []
    for cname in ("blosclz", "lz4", "lz4hc", "zlib", "zstd"):
        for shuffle in (0, 1, 2):
            cases.append(dict(cname=cname, clevel=5, shuffle=shuffle, typesize=2, kind="brick", seed=1, nbytes=16000, blocksize=0))
            cases.append(dict(cname=cname, clevel=3, shuffle=shuffle, typesize=2, kind="brick", seed=2, nbytes=33001, blocksize=8192))
            cases.append(dict(cname=cname, clevel=9, shuffle=shuffle, typesize=4, kind="f32", seed=3, nbytes=12000, blocksize=4096))
        cases.append(dict(cname=cname, clevel=5, shuffle=1, typesize=8, kind="runs", seed=4, nbytes=40000, blocksize=16384))
        cases.append(dict(cname=cname, clevel=5, shuffle=2, typesize=1, kind="runs", seed=5, nbytes=20000, blocksize=0))
        cases.append(dict(cname=cname, clevel=5, shuffle=2, typesize=3, kind="runs", seed=6, nbytes=9999, blocksize=0))
        cases.append(dict(cname=cname, clevel=1, shuffle=1, typesize=2, kind="noise", seed=7, nbytes=5000, blocksize=0))
        cases.append(dict(cname=cname, clevel=5, shuffle=1, typesize=2, kind="brick", seed=8, nbytes=100, blocksize=0))
        cases.append(dict(cname=cname, clevel=5, shuffle=0, typesize=1, kind="runs", seed=9, nbytes=1, blocksize=0))
    # the production codec on a whole (small) chunk: (1, 1, 8, 64, 64) uint16
    cases.append(dict(cname="zstd", clevel=3, shuffle=1, typesize=2, kind="brick", seed=10, nbytes=8 * 64 * 64 * 2, blocksize=0))
    cases.append(dict(cname="lz4", clevel=5, shuffle=1, typesize=2, kind="brick", seed=11, nbytes=8 * 64 * 64 * 2, blocksize=0))

    out = {"blosc_version": np.array(version)}
    total = 0
    for i, c in enumerate(cases):
        raw = payload(c["kind"], c["seed"], c["nbytes"])
        cap = len(raw) + 16
        buf = ctypes.create_string_buffer(cap)
        n = lib.blosc_compress_ctx(c["clevel"], c["shuffle"], c["typesize"], len(raw), raw, buf, cap, c["cname"].encode(),
                                   c["blocksize"], 1)
        assert n > 0, (c, n)
        frame = buf.raw[:n]
        back = ctypes.create_string_buffer(max(len(raw), 1))
        assert lib.blosc_decompress_ctx(frame, back, len(raw), 1) == len(raw) and back.raw[: len(raw)] == raw
        out["frame_%03d" % i] = np.frombuffer(frame, np.uint8)
        out["case_%03d" % i] = np.array("%(cname)s %(clevel)d %(shuffle)d %(typesize)d %(kind)s %(seed)d %(nbytes)d %(blocksize)d" % c)
        total += n
        print(i, c, "->", n, "bytes, flags 0x%02x, blocksize %d" % (frame[2], int.from_bytes(frame[8:12], "little")))
    np.savez(OUT, **out)
    print("c-blosc", version, ":", len(cases), "frames,", total, "bytes ->", OUT)


if __name__ == "__main__":
    main()
