"""Blosc chunks (the production codec: ``Blosc(cname="zstd", clevel=3, shuffle=SHUFFLE)``, reference
``zarr_destriper.py:1066-1074``) through the native container code of ``csrc/dsx_io.h``.

PARITY PINNED (round 3): the image carries the real c-blosc 1.21.0 (``/opt/conda/lib/libblosc.so.1``, the library
numcodecs.Blosc wraps; numcodecs itself is absent).  ``oracle/make_golden_blosc.py`` wrote ``tests/golden/blosc_frames.npz``
with it -- 77 frames: blosclz / lz4 / lz4hc / zlib / zstd inside, no / byte / bit shuffle, type sizes 1 ... 8, split and
unsplit blocks, short last blocks, stored frames -- which the native reader must decode to the seeded payloads; and where
the library is present (this container, the GPU box) the native writer's frames are decoded by the real
``blosc_decompress_ctx``.  Besides: a second, independent reading of the published c-blosc 1.x frame format in pure Python
below (``py_blosc_*``; inner streams through Python's ``zlib`` and, via ctypes, the image's ``libzstd.so.1``).  Frames
assembled by the Python writer -- with split blocks, leftover blocks, stored streams and zlib inside -- must decode
natively, and frames of the native writer must decode with the Python reader.  All host code: no GPU needed.
"""

import ctypes
import json
import os
import struct
import zlib

import numpy as np
import pytest

from aind_smartspim_destripe_amd import engine as eng_mod
from aind_smartspim_destripe_amd import mini_zarr
from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

SHUFFLE, MEMCPYED, DONT_SPLIT = 0x1, 0x2, 0x10
ZLIB, ZSTD = 3, 4


def _zstd():
    lib = ctypes.CDLL("libzstd.so.1")
    lib.ZSTD_compress.restype = ctypes.c_size_t
    lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    lib.ZSTD_decompress.restype = ctypes.c_size_t
    lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.ZSTD_compressBound.restype = ctypes.c_size_t
    lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
    return lib


def _inner_compress(codec, raw):
    if codec == ZLIB:
        return zlib.compress(raw, 5)
    lib = _zstd()
    cap = lib.ZSTD_compressBound(len(raw))
    buf = ctypes.create_string_buffer(cap)
    n = lib.ZSTD_compress(buf, cap, raw, len(raw), 3)
    assert not lib.ZSTD_isError(n)
    return buf.raw[:n]


def _inner_decompress(codec, comp, n):
    if codec == ZLIB:
        return zlib.decompress(comp)
    lib = _zstd()
    buf = ctypes.create_string_buffer(n)
    got = lib.ZSTD_decompress(buf, n, comp, len(comp))
    assert not lib.ZSTD_isError(got) and got == n
    return buf.raw


def _shuffle(block, typesize):
    ne = len(block) // typesize
    body = np.frombuffer(block[: ne * typesize], np.uint8).reshape(ne, typesize).T.tobytes()
    return body + block[ne * typesize :]


def _unshuffle(block, typesize):
    ne = len(block) // typesize
    body = np.frombuffer(block[: ne * typesize], np.uint8).reshape(typesize, ne).T.tobytes()
    return body + block[ne * typesize :]


def py_blosc_write(raw, typesize, blocksize, codec, shuffle, split):
    """c-blosc 1.x frame, written from the format description: header (version 2, codec format 1, flags, typesize,
    nbytes, blocksize, cbytes), block start table, per block `nsplits` streams of (int32 length, bytes); a stream as
    long as its uncompressed size is stored as is."""
    nblocks = -(-len(raw) // blocksize)
    flags = (SHUFFLE if shuffle and typesize > 1 else 0) | (0 if split else DONT_SPLIT) | (codec << 5)
    body, starts = b"", []
    for b in range(nblocks):
        blk = raw[b * blocksize : (b + 1) * blocksize]
        leftover = len(blk) != blocksize
        if flags & SHUFFLE:
            blk = _shuffle(blk, typesize)
        nsplits = typesize if (split and not leftover and typesize <= 16 and blocksize // typesize >= 128) else 1
        ne = len(blk) // nsplits
        starts.append(16 + 4 * nblocks + len(body))
        for j in range(nsplits):
            part = blk[j * ne : (j + 1) * ne]
            comp = _inner_compress(codec, part)
            if len(comp) >= len(part):
                comp = part
            body += struct.pack("<i", len(comp)) + comp
    table = b"".join(struct.pack("<i", x) for x in starts)
    head = struct.pack("<BBBBIII", 2, 1, flags, typesize, len(raw), blocksize, 16 + len(table) + len(body))
    return head + table + body


def py_blosc_read(frame):
    version, _, flags, typesize, nbytes, blocksize, cbytes = struct.unpack("<BBBBIII", frame[:16])
    assert version == 2 and cbytes == len(frame)
    if flags & MEMCPYED:
        return frame[16 : 16 + nbytes]
    codec = flags >> 5
    nblocks = -(-nbytes // blocksize)
    out = b""
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        nsplits = typesize if (not (flags & DONT_SPLIT) and not leftover and typesize <= 16 and blocksize // typesize >= 128) else 1
        ne = bsize // nsplits
        pos = struct.unpack("<i", frame[16 + 4 * b : 20 + 4 * b])[0]
        blk = b""
        for _ in range(nsplits):
            cs = struct.unpack("<i", frame[pos : pos + 4])[0]
            part = frame[pos + 4 : pos + 4 + cs]
            blk += part if cs == ne else _inner_decompress(codec, part, ne)
            pos += 4 + cs
        out += _unshuffle(blk, typesize) if (flags & SHUFFLE and typesize > 1) else blk
    return out


def _brick(seed=0, shape=(64, 128, 128)):
    rs = np.random.RandomState(seed)
    return (rs.poisson(150, shape) * np.exp(0.15 * rs.randn(shape[1]))[None, :, None]).astype(np.uint16)


def test_native_writer_frames_read_back_and_by_the_independent_reader():
    brick = _brick()
    raw = brick.tobytes()
    frame = mini_zarr.blosc_encode(raw, 2, clevel=3, shuffle=True)
    version, vlz, flags, typesize, nbytes, blocksize, cbytes = struct.unpack("<BBBBIII", frame[:16])
    assert (version, vlz, typesize, nbytes, cbytes) == (2, 1, 2, len(raw), len(frame))
    assert flags == SHUFFLE | DONT_SPLIT | (ZSTD << 5) and blocksize == 256 * 1024
    assert len(frame) < 0.8 * len(raw)  # Poisson(150) * stripes: the high bytes compress away
    assert mini_zarr.blosc_decode(frame, len(raw)) == raw
    assert py_blosc_read(frame) == raw
    # sizes around the block / element boundaries, 4-byte elements, no shuffle, level 9
    for n, ts, sh, lvl in ((0, 2, True, 3), (1, 2, True, 3), (127, 2, True, 3), (128, 2, True, 3), (129, 2, True, 3),
                           (256 * 1024 + 2, 2, True, 3), (3 * 256 * 1024 + 7, 4, True, 1), (70001, 4, False, 9),
                           (4096, 1, True, 5)):  # fmt: skip
        data = (np.arange(n, dtype=np.uint32) // 3).astype(np.uint8).tobytes()
        f = mini_zarr.blosc_encode(data, ts, clevel=lvl, shuffle=sh)
        assert len(f) <= n + 16
        assert mini_zarr.blosc_decode(f, n) == data, (n, ts, sh, lvl)
        assert py_blosc_read(f) == data, (n, ts, sh, lvl)
    # incompressible data is stored (memcpyed frame), clevel 0 as well
    noise = np.random.RandomState(1).bytes(100000)
    for lvl in (3, 0):
        f = mini_zarr.blosc_encode(noise, 2, clevel=lvl, shuffle=True)
        assert f[2] & MEMCPYED and len(f) == len(noise) + 16
        assert mini_zarr.blosc_decode(f, len(noise)) == noise and py_blosc_read(f) == noise


@pytest.mark.parametrize("codec", [ZLIB, ZSTD])
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("shuffle", [False, True])
def test_frames_assembled_from_the_format_description_decode_natively(codec, split, shuffle):
    """Block sizes c-blosc may pick (not this writer's), split streams per byte plane, a leftover block, stored streams."""
    raw = _brick(2, (8, 64, 64)).tobytes()
    for typesize, blocksize, n in ((2, 32768, len(raw)), (2, 8192, len(raw) - 1000), (4, 4096, 50000), (2, 200, 1000)):
        data = raw[:n]
        frame = py_blosc_write(data, typesize, blocksize, codec, shuffle, split)
        assert py_blosc_read(frame) == data
        assert mini_zarr.blosc_decode(frame, n) == data, (typesize, blocksize, n)
    # a block of noise inside compressible data: its stream is stored as is (length == block size)
    mixed = raw[:32768] + np.random.RandomState(3).bytes(32768) + raw[:32768]
    frame = py_blosc_write(mixed, 2, 32768, codec, shuffle, split)
    assert mini_zarr.blosc_decode(frame, len(mixed)) == mixed


def payload(kind, seed, nbytes):
    """The payloads of oracle/make_golden_blosc.py, regenerated from their seeds."""
    rs = np.random.RandomState(seed)
    if kind == "brick":
        n = nbytes // 2
        b = ((np.arange(n) % 977) * 13 + rs.randint(0, 40, n) + 300).astype("<u2").tobytes()
    elif kind == "noise":
        b = rs.bytes(nbytes)
    elif kind == "runs":
        b = np.repeat(rs.randint(0, 256, nbytes // 37 + 1).astype(np.uint8), 37).tobytes()
    elif kind == "f32":
        b = np.cumsum(rs.standard_normal(nbytes // 4 + 1).astype(np.float32)).astype("<f4").tobytes()
    else:
        raise ValueError(kind)
    return (b + bytes(nbytes))[:nbytes]


def test_frames_written_by_the_real_c_blosc_decode_natively():
    """tests/golden/blosc_frames.npz (c-blosc 1.21.0, oracle/make_golden_blosc.py): every inner codec, shuffle mode,
    split layout and type size it produced; the stored / split / bit-shuffled / blosclz paths must all have been seen."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "blosc_frames.npz"))
    assert str(g["blosc_version"]) == "1.21.0"
    seen = set()
    n = 0
    while "frame_%03d" % n in g.files:
        frame = g["frame_%03d" % n].tobytes()
        cname, clevel, shuffle, typesize, kind, seed, nbytes, blocksize = str(g["case_%03d" % n]).split()
        raw = payload(kind, int(seed), int(nbytes))
        assert mini_zarr.blosc_decode(frame, len(raw)) == raw, str(g["case_%03d" % n])
        flags = frame[2]
        seen.add((flags >> 5, "stored" if flags & MEMCPYED else "bit" if flags & 0x4 else "byte" if flags & SHUFFLE else "plain",
                  bool(flags & DONT_SPLIT)))
        n += 1
    assert n == 77
    assert {c for c, _, _ in seen} == {0, 1, 3, 4}                       # blosclz, lz4 (+ lz4hc), zlib, zstd
    assert {m for _, m, _ in seen} == {"stored", "bit", "byte", "plain"}
    assert {d for _, _, d in seen} == {False, True}                      # split and unsplit block streams


REAL_BLOSC = "/opt/conda/lib/libblosc.so.1"


@pytest.mark.skipif(not os.path.exists(REAL_BLOSC), reason="the image's c-blosc is not here")
def test_native_writer_frames_decode_with_the_real_c_blosc():
    """What the chunk writer stores must be readable by the library the reference (numcodecs) reads with -- and what
    that library writes for the production settings by this reader, on a whole production-size chunk."""
    lib = ctypes.CDLL(REAL_BLOSC)
    lib.blosc_decompress_ctx.restype = ctypes.c_int
    lib.blosc_decompress_ctx.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.blosc_cbuffer_validate.restype = ctypes.c_int
    lib.blosc_cbuffer_validate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    lib.blosc_compress_ctx.restype = ctypes.c_int
    lib.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cases = [(_brick(7, (64, 128, 128)).tobytes(), 2, 3, True)]          # the production chunk (1, 1, 64, 128, 128) uint16
    cases += [(payload("brick", 20, n), 2, 3, True) for n in (0, 1, 2, 127, 128, 129, 256 * 1024, 256 * 1024 + 2, 600001)]
    cases += [(payload("f32", 21, 300000), 4, 9, True), (payload("runs", 22, 70001), 1, 5, True),
              (payload("runs", 23, 70001), 8, 1, False), (payload("noise", 24, 50000), 2, 3, True),
              (payload("brick", 25, 50000), 2, 0, True)]
    for raw, typesize, clevel, shuffle in cases:
        frame = mini_zarr.blosc_encode(raw, typesize, clevel=clevel, shuffle=shuffle)
        nb = ctypes.c_size_t(0)
        assert lib.blosc_cbuffer_validate(frame, len(frame), ctypes.byref(nb)) == 0 and nb.value == len(raw)
        back = ctypes.create_string_buffer(max(len(raw), 1))
        assert lib.blosc_decompress_ctx(frame, back, len(raw), 1) == len(raw), (len(raw), typesize, clevel, shuffle)
        assert back.raw[: len(raw)] == raw
    # the other direction at production size and settings: Blosc(cname="zstd", clevel=3, shuffle=SHUFFLE)
    raw = cases[0][0]
    buf = ctypes.create_string_buffer(len(raw) + 16)
    n = lib.blosc_compress_ctx(3, 1, 2, len(raw), raw, buf, len(raw) + 16, b"zstd", 0, 1)
    assert n > 0 and mini_zarr.blosc_decode(buf.raw[:n], len(raw)) == raw
    # (and the sizes agree to a few percent: same inner codec and level, other block size)
    ours = len(mini_zarr.blosc_encode(raw, 2, clevel=3, shuffle=True))
    assert abs(ours - n) <= 0.05 * n, (ours, n)


def test_malformed_frames_fail_loudly():
    raw = _brick(4, (4, 64, 64)).tobytes()
    good = mini_zarr.blosc_encode(raw, 2)
    for bad, what in (
        (good[:10], "shorter than its header"),
        (good[:-5], "does not fit the file"),
        (good, "chunk needs"),  # decoded into a chunk of another size
        (bytes([9]) + good[1:], "format version"),
        (good[:2] + bytes([good[2] & 0x1F]) + good[3:], "blosclz"),  # inner codec 0
        (good[:2] + bytes([(good[2] & 0x1F) | (2 << 5)]) + good[3:], "snappy"),  # (c-blosc 1.21 is built without it too)
        (good[:16] + struct.pack("<i", len(good) + 100) + good[20:], "outside the frame"),
        (good[:8] + struct.pack("<I", 0) + good[12:], "block size"),
    ):
        with pytest.raises(ValueError, match=what):
            mini_zarr.blosc_decode(bad, len(raw) + (2 if what == "chunk needs" else 0))
    # a corrupted zstd stream
    broken = bytearray(good)
    broken[len(good) // 2] ^= 0xFF
    broken[len(good) // 2 + 1] ^= 0xFF
    try:
        out = mini_zarr.blosc_decode(bytes(broken), len(raw))
        assert out != raw  # zstd frames carry no checksum by default: then the damage must at least show
    except ValueError as e:
        assert "zstd" in str(e)


def test_blosc_store_roundtrip_metadata_and_native_chunk_io(tmp_path):
    vol = _brick(5, (12, 40, 56))
    a = MiniZarrArray.create(str(tmp_path / "a.zarr"), (1, 1) + vol.shape, (1, 1, 8, 16, 16), np.uint16, compressor="blosc")
    meta = json.load(open(tmp_path / "a.zarr" / ".zarray"))
    assert meta["compressor"] == {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 1, "blocksize": 0}
    a[0, 0] = vol
    b = MiniZarrArray.open(str(tmp_path / "a.zarr"))
    np.testing.assert_array_equal(b[0, 0], vol)
    np.testing.assert_array_equal(b[0, 0, 3:11, 5:33, 7:50], vol[3:11, 5:33, 7:50])
    b[0, 0, 2:5, 10:20, 10:20] = 7  # partial chunks: read - modify - write through the codec
    vol[2:5, 10:20, 10:20] = 7
    np.testing.assert_array_equal(MiniZarrArray.open(str(tmp_path / "a.zarr"))[0, 0], vol)
    first = open(a._chunk_path((0, 0, 0, 0, 0)), "rb").read()
    assert first[0] == 2 and (first[2] >> 5) == ZSTD and first[3] == 2
    # the native thread pool reads what the Python path wrote and vice versa
    lib = eng_mod.load_library()
    idxs = [(0, 0, z, y, x) for z in range(2) for y in range(3) for x in range(4)]
    got = [np.empty((8, 16, 16), np.uint16) for _ in idxs]
    n = len(idxs)

    def arrays(fn, paths, arrs, *extra):
        cp = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        dp = (ctypes.c_void_p * n)(*[x.ctypes.data for x in arrs])
        nb = (ctypes.c_size_t * n)(*[x.nbytes for x in arrs])
        return fn(None, cp, dp, nb, n, 4, *extra)

    assert arrays(lib.dsx_io_read_chunks, [a._chunk_path(i) for i in idxs], got, 2, 0) == 0
    padded = np.zeros((16, 48, 64), np.uint16)
    padded[:12, :40, :56] = vol
    for (_, _, z, y, x), g in zip(idxs, got):
        np.testing.assert_array_equal(g, padded[8 * z : 8 * z + 8, 16 * y : 16 * y + 16, 16 * x : 16 * x + 16])
    c = MiniZarrArray.create(str(tmp_path / "c.zarr"), (1, 1, 16, 48, 64), (1, 1, 8, 16, 16), np.uint16, compressor="blosc")
    assert arrays(lib.dsx_io_write_chunks_blosc, [c._chunk_path(i) for i in idxs], got, 3, 2, 1) == 0
    np.testing.assert_array_equal(c[0, 0], padded)
    # a store whose metadata names another inner codec is readable when its frames are (the frame header decides)
    d = MiniZarrArray.create(str(tmp_path / "d.zarr"), (4, 64), (4, 64), np.uint16,
                             compressor={"id": "blosc", "cname": "zlib", "clevel": 5, "shuffle": 1, "blocksize": 0})
    data = _brick(6, (1, 4, 64))[0]
    os.makedirs(os.path.dirname(d._chunk_path((0, 0))), exist_ok=True)
    open(d._chunk_path((0, 0)), "wb").write(py_blosc_write(data.tobytes(), 2, 512, ZLIB, True, True))
    np.testing.assert_array_equal(MiniZarrArray.open(str(tmp_path / "d.zarr"))[...], data)
    with pytest.raises(NotImplementedError, match="read-only"):
        d[...] = data
    # Zarr's own default codec -- Blosc(cname="lz4", clevel=5, shuffle=SHUFFLE) -- and a blosclz / bit-shuffle store, with
    # chunk files written by the real c-blosc (golden frames 76 and 6: 8 x 64 x 64 and 8 000 uint16), through both readers
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "blosc_frames.npz"))
    for k, cname, shuffle, shape in ((76, "lz4", 1, (8, 64, 64)), (6, "blosclz", 2, (8000,))):
        case = str(g["case_%03d" % k]).split()
        assert (case[0], int(case[2])) == (cname, shuffle)
        want = np.frombuffer(payload(case[4], int(case[5]), int(case[6])), "<u2").reshape(shape)
        e = MiniZarrArray.create(str(tmp_path / "e{}.zarr".format(k)), shape, shape, np.uint16,
                                 compressor={"id": "blosc", "cname": cname, "clevel": 5, "shuffle": shuffle, "blocksize": 0})
        idx = (0,) * len(shape)
        os.makedirs(os.path.dirname(e._chunk_path(idx)), exist_ok=True)
        open(e._chunk_path(idx), "wb").write(g["frame_%03d" % k].tobytes())
        np.testing.assert_array_equal(MiniZarrArray.open(e.path)[...], want)
        got1 = np.empty(shape, np.uint16)
        cp = (ctypes.c_char_p * 1)(os.fsencode(e._chunk_path(idx)))
        dp = (ctypes.c_void_p * 1)(got1.ctypes.data)
        nb = (ctypes.c_size_t * 1)(got1.nbytes)
        assert lib.dsx_io_read_chunks(None, cp, dp, nb, 1, 1, 2, 0) == 0
        np.testing.assert_array_equal(got1, want)


@pytest.mark.gpu
def test_gpu_chunk_map_on_a_production_style_blosc_store(tmp_path):
    """destripe_zarr, Blosc-zstd in and out (input chunks as the production acquisition writes them, output codec of
    zarr_destriper.py:1066-1074), overlapped device path and host path: same voxels as the zlib store run, and every
    output chunk file is a Blosc frame."""
    from aind_smartspim_destripe_amd import synth
    from aind_smartspim_destripe_amd import zarr_destriper as zd

    stack = synth.synthetic_stack(20, 96, 128, n_unique=5)
    outs = {}
    for comp in ("zlib", "blosc"):
        src = MiniZarrArray.create(str(tmp_path / "in_{}.zarr".format(comp)), (1, 1) + stack.shape, (1, 1, 8, 32, 32),
                                   np.uint16, compressor=comp)  # fmt: skip
        src[0, 0] = stack
        for mode in (False, True):
            path = str(tmp_path / "out_{}_{}.zarr".format(comp, int(mode)))
            n, _ = zd.destripe_zarr_store(str(tmp_path / "in_{}.zarr".format(comp)), path, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG,
                                    prediction_chunksize=(8, 96, 128), output_chunks=(1, 1, 8, 32, 32), device=0,
                                    device_retile=mode, compressor=comp)  # fmt: skip
            assert n == 20
            outs[comp, mode] = MiniZarrArray.open(path)[0, 0]
    assert outs["blosc", True].any()
    for key in (("blosc", False), ("blosc", True), ("zlib", True)):
        np.testing.assert_array_equal(outs[key], outs["zlib", False])
    out = MiniZarrArray.open(str(tmp_path / "out_blosc_1.zarr"))
    assert json.load(open(os.path.join(out.path, ".zarray")))["compressor"]["cname"] == "zstd"
    frame = open(out._chunk_path((0, 0, 1, 2, 3)), "rb").read()
    assert frame[0] == 2 and (frame[2] >> 5) == ZSTD and struct.unpack("<I", frame[4:8])[0] == 8 * 32 * 32 * 2


def test_mutated_frames_never_leave_their_buffers(tmp_path):
    """The decoders of csrc/dsx_io.h (Blosc container, PNG scanline filters) take file contents: 20 000 random corruptions
    and truncations of a good frame under AddressSanitizer / UBSan (host build; tests/host/codec_mutation_check.cpp)."""
    import subprocess

    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "codec_mutation_check.cpp")
    exe = str(tmp_path / "codec_mutation_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-o", exe, src, "-lz", "-lpthread", "-ldl"], check=True, cwd=os.path.dirname(src))
    out = subprocess.run([exe, "20000"], check=True, capture_output=True, text=True, cwd=os.path.dirname(src)).stdout
    assert out.strip().endswith("ok") and "rejected" in out, out
