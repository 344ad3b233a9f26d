"""Rows f1 / f3 of SURVEY section 8: Zarr brick re-tiling and the 2x2x2 windowed-mean pyramid.

CPU: the NumPy restatements in ``oracle/format_oracle.py`` against the Zarr-v2 store layout and
hand-computed values.  GPU (``-m gpu``): the HIP kernels through the C ABI, bit-exact against the
restatements (integer / byte work), and the device re-tiling path of ``destripe_zarr`` against the host
gather / scatter path.
"""

import os

import numpy as np
import pytest

from aind_smartspim_destripe_amd import synth
from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray
from oracle import format_oracle as fo

SHAPES = [
    # (Z, H, W), brick, z0
    ((64, 256, 384), (64, 128, 128), 0),
    ((10, 100, 120), (4, 32, 32), 0),  # partial bricks on every axis
    ((7, 33, 50), (4, 16, 20), 3),  # stack starts inside a brick, 4-byte vectors
    ((5, 17, 35), (2, 8, 7), 1),  # odd widths: scalar path
    ((3, 40, 44), (8, 64, 64), 0),  # bricks larger than the stack
]


def _vol(shape, seed=0):
    return np.random.RandomState(seed).randint(0, 65536, shape).astype(np.uint16)


# ---------------------------------------------------------------------------------------------- CPU


@pytest.mark.parametrize("zyx,brick,z0", SHAPES)
def test_brick_oracle_roundtrip(zyx, brick, z0):
    v = _vol(zyx)
    b = fo.planes_to_bricks(v, brick, z0)
    assert b.shape == fo.brick_grid(zyx, brick, z0) + tuple(brick)
    np.testing.assert_array_equal(fo.bricks_to_planes(b, zyx, z0), v)
    assert int(b.astype(np.uint64).sum()) == int(v.astype(np.uint64).sum())  # padding is the fill value 0


def test_brick_oracle_is_the_zarr_chunk_layout(tmp_path):
    """Brick (bz, by, bx) of the restatement == bytes of chunk file bz/by/bx of a Zarr-v2 store."""
    v = _vol((10, 100, 120), 1)
    a = MiniZarrArray.create(str(tmp_path / "a.zarr"), (1, 1) + v.shape, (1, 1, 4, 32, 32), np.uint16)
    a[0, 0] = v
    b = fo.planes_to_bricks(v, (4, 32, 32))
    for idx in [(0, 0, 0), (2, 3, 3), (1, 2, 0)]:
        raw = np.fromfile(os.path.join(str(tmp_path / "a.zarr"), "0", "0", *map(str, idx)), dtype="<u2")
        np.testing.assert_array_equal(raw, b[idx].ravel())
        flat = np.empty(4 * 32 * 32, np.uint16)
        a.read_chunk_into((0, 0) + idx, flat)
        np.testing.assert_array_equal(flat, b[idx].ravel())
    # write_chunk_flat is the inverse
    c = MiniZarrArray.create(str(tmp_path / "c.zarr"), (1, 1) + v.shape, (1, 1, 4, 32, 32), np.uint16, compressor="zlib")
    for idx in np.ndindex(*b.shape[:3]):
        c.write_chunk_flat((0, 0) + idx, b[idx].reshape(c.chunks))
    np.testing.assert_array_equal(MiniZarrArray.open(str(tmp_path / "c.zarr"))[0, 0], v)


def test_windowed_mean_oracle_values():
    v = np.zeros((2, 2, 4), np.uint16)
    v[..., :2] = [[[1, 2], [3, 4]], [[5, 6], [7, 9]]]  # sum 37 -> 4.625 -> 4 (truncation, not rounding)
    v[..., 2:] = 65535  # no overflow: the mean is taken in float64
    np.testing.assert_array_equal(fo.windowed_mean_u16(v), [[[4, 65535]]])
    # odd trailing voxels are cropped; every level derives from the previous one
    p = fo.pyramid(_vol((9, 21, 35), 2), 3)
    assert [a.shape for a in p] == [(9, 21, 35), (4, 10, 17), (2, 5, 8)]
    a = p[0].astype(np.uint32)
    s = sum(a[dz:8:2, dy:20:2, dx:34:2] for dz in (0, 1) for dy in (0, 1) for dx in (0, 1))
    np.testing.assert_array_equal(p[1], (s >> 3).astype(np.uint16))


def test_pyramid_argument_checks():
    from aind_smartspim_destripe_amd import pyramid

    with pytest.raises(ValueError):
        pyramid.compute_pyramid(np.zeros((4, 4, 4), np.uint16), 2, (2, 2, 1))
    with pytest.raises(ValueError):
        pyramid.compute_pyramid(np.zeros((4, 4, 4), np.float32), 2, (2, 2, 2))
    with pytest.raises(ValueError):
        pyramid.compute_pyramid(np.zeros((2, 4, 4, 4), np.uint16), 2, (1, 2, 2, 2))


# ---------------------------------------------------------------------------------------------- GPU


@pytest.fixture(scope="module")
def eng():
    from aind_smartspim_destripe_amd import engine

    e = engine.DestripeEngine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("zyx,brick,z0", SHAPES)
def test_gpu_brick_retile_bit_exact(eng, zyx, brick, z0):
    v = _vol(zyx, 3)
    want = fo.planes_to_bricks(v, brick, z0)
    d_planes, d_bricks, d_back = eng.alloc(v.nbytes), eng.alloc(want.nbytes), eng.alloc(v.nbytes)
    try:
        d_planes.upload(v)
        d_bricks.upload(np.full(want.shape, 0xABCD, np.uint16))  # every brick voxel must be written
        eng.planes_to_bricks(d_planes, d_bricks, zyx, brick, z0)
        eng.sync()
        got = d_bricks.download(want.shape, np.uint16)
        np.testing.assert_array_equal(got, want)
        eng.bricks_to_planes(d_bricks, d_back, zyx, brick, z0)
        eng.sync()
        np.testing.assert_array_equal(d_back.download(zyx, np.uint16), v)
    finally:
        for b in (d_planes, d_bricks, d_back):
            b.free()


@pytest.mark.gpu
def test_gpu_brick_retile_rejects_bad_shapes(eng):
    from aind_smartspim_destripe_amd.engine import DsxError

    d = eng.alloc(1024)
    try:
        with pytest.raises(DsxError):
            eng.bricks_to_planes(d, d, (0, 4, 4), (2, 2, 2))
        with pytest.raises(DsxError):
            eng.planes_to_bricks(d, d, (4, 4, 4), (2, 0, 2))
        with pytest.raises(ValueError):
            eng.downsample2(d, d, (1, 4, 4))
    finally:
        d.free()


@pytest.mark.gpu
@pytest.mark.parametrize("zyx", [(64, 256, 512), (9, 21, 35), (2, 2, 2), (5, 64, 40), (4, 6, 2050)])
def test_gpu_downsample_bit_exact(eng, zyx):
    v = _vol(zyx, 4)
    v[0, :2, :8] = 65535  # saturated windows must not overflow
    want = fo.windowed_mean_u16(v)
    d_src, d_dst = eng.alloc(v.nbytes), eng.alloc(max(want.nbytes, 16))
    try:
        d_src.upload(v)
        eng.downsample2(d_src, d_dst, zyx)
        eng.sync()
        np.testing.assert_array_equal(d_dst.download(want.shape, np.uint16), want)
    finally:
        d_src.free()
        d_dst.free()


@pytest.mark.gpu
def test_gpu_compute_pyramid_and_multiscale(tmp_path):
    from aind_smartspim_destripe_amd import pyramid

    v = _vol((1, 1, 70, 130, 200), 5)
    levels = pyramid.compute_pyramid(v, 3, [1, 1, 2, 2, 2])
    want = fo.pyramid(v[0, 0], 3)
    assert [a.shape for a in levels] == [(1, 1) + w.shape for w in want]
    for a, w in zip(levels, want):
        np.testing.assert_array_equal(a[0, 0], w)
    # more levels than the volume supports: stops when an axis drops below 2
    assert len(pyramid.compute_pyramid(_vol((2, 8, 8)), 5, (2, 2, 2))) == 2
    # store-to-store driver
    src = MiniZarrArray.create(str(tmp_path / "g" / "0"), v.shape, (1, 1, 64, 128, 128), np.uint16, compressor="zlib")
    src[...] = v
    shapes = pyramid.write_pyramid_levels(str(tmp_path / "g" / "0"), str(tmp_path / "g"), n_levels=3, compressor="zlib")
    assert shapes == [(1, 1, 35, 65, 100), (1, 1, 17, 32, 50)]
    for i in (1, 2):
        np.testing.assert_array_equal(MiniZarrArray.open(str(tmp_path / "g" / str(i)))[0, 0], want[i])


@pytest.mark.gpu
@pytest.mark.parametrize("in_chunks", [(1, 1, 8, 32, 32), (1, 1, 5, 48, 64)])
def test_destripe_zarr_device_retile_equals_host_path(tmp_path, in_chunks):
    """Row f1 end to end: the device brick path writes byte-identical chunks to the host gather / scatter path."""
    from aind_smartspim_destripe_amd import zarr_destriper as zd

    stack = synth.synthetic_stack(20, 96, 128, n_unique=5)
    src = MiniZarrArray.create(str(tmp_path / "X_0_Y_0.zarr"), (1, 1) + stack.shape, in_chunks, np.uint16,
                               compressor="zlib")  # fmt: skip
    src[0, 0] = stack
    outs = {}
    for mode in (False, True):
        path = str(tmp_path / "out_{}.zarr".format(int(mode)))
        total = 0
        for rank in range(2):
            n, _ = zd.destripe_zarr_store(str(tmp_path / "X_0_Y_0.zarr"), path, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG,
                                    prediction_chunksize=(8, 96, 128), output_chunks=(1, 1, 8, 32, 32), rank=rank,
                                    world_size=2, device=0, device_retile=mode)  # fmt: skip
            total += n
        assert total == 20
        outs[mode] = MiniZarrArray.open(path)[0, 0]
    np.testing.assert_array_equal(outs[True], outs[False])
    assert outs[True].any()
    # the partial last z-chunk is stored full-sized with the fill value behind the data
    raw = np.empty(8 * 32 * 32, np.uint16)
    MiniZarrArray.open(str(tmp_path / "out_1.zarr")).read_chunk_into((0, 0, 2, 0, 0), raw)
    assert raw.reshape(8, 32, 32)[4:].sum() == 0 and raw.reshape(8, 32, 32)[:4].any()


def test_native_chunk_io_roundtrip(tmp_path):
    """dsx_io_read_chunks / dsx_io_write_chunks (native threads, no GPU): raw and zlib chunks agree with the
    Python store code in both directions, a missing chunk reads as the fill value, a bad path fails loudly."""
    import ctypes

    from aind_smartspim_destripe_amd import engine as eng_mod
    from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

    lib = eng_mod.load_library()

    def call(fn, paths, arrays, *extra):
        n = len(paths)
        cp = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        dp = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrays])
        nb = (ctypes.c_size_t * n)(*[a.nbytes for a in arrays])
        return fn(None, cp, dp, nb, n, 4, *extra)

    rs = np.random.RandomState(5)
    vol = rs.randint(0, 65535, (8, 32, 48)).astype(np.uint16)
    for comp, level in ((None, -1), ("zlib", 1)):
        a = MiniZarrArray.create(str(tmp_path / "a_{}".format(comp)), (1, 1, 8, 32, 48), (1, 1, 4, 16, 16), np.uint16,
                                 compressor=comp)  # fmt: skip
        idxs = [(0, 0, z, y, x) for z in range(2) for y in range(2) for x in range(3)]
        bricks = [np.ascontiguousarray(vol[4 * z : 4 * z + 4, 16 * y : 16 * y + 16, 16 * x : 16 * x + 16]) for _, _, z, y, x in idxs]
        assert call(lib.dsx_io_write_chunks, [a._chunk_path(i) for i in idxs], bricks, level) == 0
        np.testing.assert_array_equal(a[0, 0], vol)  # the Python reader understands what the native writer wrote
        b = MiniZarrArray.create(str(tmp_path / "b_{}".format(comp)), (1, 1, 8, 32, 48), (1, 1, 4, 16, 16), np.uint16,
                                 compressor=comp, fill_value=7)  # fmt: skip
        b[0, 0, :, :, :32] = vol[:, :, :32]  # chunks x == 2 are never written
        got = [np.empty((4, 16, 16), np.uint16) for _ in idxs]
        assert call(lib.dsx_io_read_chunks, [b._chunk_path(i) for i in idxs], got, 0 if comp is None else 1, 7) == 0
        for (_, _, z, y, x), g in zip(idxs, got):
            want = vol[4 * z : 4 * z + 4, 16 * y : 16 * y + 16, 16 * x : 16 * x + 16] if x < 2 else np.full((4, 16, 16), 7)
            np.testing.assert_array_equal(g, want)
    bad = [np.zeros(8, np.uint16)]
    open(str(tmp_path / "short"), "wb").write(b"abc")
    assert call(lib.dsx_io_read_chunks, [str(tmp_path / "short")], bad, 0, 0) == -7
    assert b"short" in lib.dsx_last_error(None)
