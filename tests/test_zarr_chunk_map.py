"""Chunk map (execute_worker / destripe_zarr mirror) and the minimal Zarr-v2 store.

CPU tests patch the GPU call, as the reference's own tests patch the filter
(code/tests/test_filtering.py:242-281); the GPU test runs the whole map on a small synthetic tile
and checks every plane against the CPU oracle.
"""

import json
import logging
import os
from unittest.mock import patch

import numpy as np
import pytest

from aind_smartspim_destripe_amd import synth, zarr_destriper as zd
from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

LOG = logging.getLogger("test")


@pytest.mark.parametrize("compressor,sep", [(None, "/"), ("zlib", "."), ("zlib", "/")])
def test_mini_zarr_roundtrip(tmp_path, compressor, sep):
    rs = np.random.RandomState(0)
    full = rs.randint(0, 65535, (1, 1, 70, 50, 45)).astype(np.uint16)
    a = MiniZarrArray.create(str(tmp_path / "a.zarr"), full.shape, (1, 1, 32, 16, 16), np.uint16,
                             compressor=compressor, dimension_separator=sep)  # fmt: skip
    a[...] = full
    b = MiniZarrArray.open(str(tmp_path / "a.zarr"))
    np.testing.assert_array_equal(b[...], full)
    np.testing.assert_array_equal(b[0, 0, 3:40, 5:33, 7], full[0, 0, 3:40, 5:33, 7])
    # partial overwrite crossing chunk borders; float values are truncated like a NumPy cast
    b[0, 0, 30:41, 10:20, 10:30] = np.full((11, 10, 20), 7.9, np.float32)
    full[0, 0, 30:41, 10:20, 10:30] = 7
    np.testing.assert_array_equal(MiniZarrArray.open(str(tmp_path / "a.zarr"))[...], full)
    meta = json.load(open(tmp_path / "a.zarr" / ".zarray"))
    assert meta["zarr_format"] == 2 and meta["chunks"] == [1, 1, 32, 16, 16]
    # never-written chunks read as fill_value
    c = MiniZarrArray.create(str(tmp_path / "c.zarr"), (4, 40, 40), (2, 16, 16), np.uint16)
    assert int(c[...].sum()) == 0
    with pytest.raises(NotImplementedError):
        MiniZarrArray(str(tmp_path), dict(meta, compressor={"id": "lzma"}))


def test_pad_and_coordinates():
    assert zd.pad_array_n_d(np.zeros((3, 3)), dim=5).shape == (1, 1, 1, 3, 3)  # reference test
    with pytest.raises(ValueError):
        zd.pad_array_n_d(np.zeros((3, 3)), dim=6)
    sc = (slice(128, 192), slice(0, 100), slice(0, 120))
    internal = [(slice(0, 64), slice(0, 100), slice(0, 120))]
    glob, starts, stops = zd.recover_global_position(sc, internal)
    assert glob == (slice(128, 192), slice(0, 100), slice(0, 120)) and starts == (128, 0, 0)
    g, l = zd.unpad_global_coords(glob, (64, 100, 120), (0, 0, 0), (1, 1, 300, 100, 120))
    assert g == glob and l == (slice(0, 64), slice(0, 100), slice(0, 120))
    g, l = zd.unpad_global_coords(glob, (64, 100, 120), (4, 0, 0), (1, 1, 300, 100, 120))
    assert g[0] == slice(132, 188) and l[0] == slice(4, 60) and l[1] == slice(0, 100)


def test_execute_worker_places_block(tmp_path):
    """Same placement as the reference: block z[64:128) of a (1,1,150,40,48) array; last block clipped."""
    out = MiniZarrArray.create(str(tmp_path / "o.zarr"), (1, 1, 150, 40, 48), (1, 1, 64, 16, 16), np.uint16)
    calls = {}

    def fake(planes, **kw):
        calls.update(kw, shape=planes.shape, dtype=planes.dtype)
        return (planes + 2).astype(np.uint16)

    data = np.arange(64 * 40 * 48, dtype=np.float32).reshape(1, 64, 40, 48) % 1000
    with patch("aind_smartspim_destripe_amd.filtering.destripe_planes", side_effect=fake):
        zd.execute_worker(data, (slice(64, 128), slice(0, 40), slice(0, 48)),
                          [(slice(0, 64), slice(0, 40), slice(0, 48))], synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG,
                          (0, 0, 0), out, None, "tile_X_0001_Y_0002.zarr", LOG)  # fmt: skip
        assert calls["microscope_high_int"] == 2500 and calls["input_tile_path"] == "tile_X_0001_Y_0002"
        assert calls["shape"] == (64, 40, 48) and calls["out_dtype"] == np.uint16
        got = out[...]
        np.testing.assert_array_equal(got[0, 0, 64:128], (data[0] + 2).astype(np.uint16))
        assert got[0, 0, :64].sum() == 0 and got[0, 0, 128:].sum() == 0
        # block that sticks out of the dataset in z is clipped (zarr_destriper.py:301-309)
        zd.execute_worker(data, (slice(128, 192), slice(0, 40), slice(0, 48)),
                          [(slice(0, 64), slice(0, 40), slice(0, 48))], synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG,
                          (0, 0, 0), out, None, "t.zarr", LOG)  # fmt: skip
        np.testing.assert_array_equal(out[0, 0, 128:150], (data[0, :22] + 2).astype(np.uint16))


def test_iter_blocks_and_plane_guard(tmp_path):
    blocks = list(zd.iter_blocks((150, 40, 48), (64, 40, 48)))
    assert [b[0][0] for b in blocks] == [slice(0, 64), slice(64, 128), slice(128, 150)]
    blocks = list(zd.iter_blocks((150, 40, 48), (64, 40, 48), z_range=(64, 128)))
    assert len(blocks) == 1 and blocks[0][0][0] == slice(64, 128)
    MiniZarrArray.create(str(tmp_path / "i.zarr"), (10, 40, 48), (4, 16, 16), np.uint16)
    with pytest.raises(ValueError):
        zd.destripe_zarr_store(str(tmp_path / "i.zarr"), str(tmp_path / "o.zarr"), synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG,
                         prediction_chunksize=(4, 20, 48))  # fmt: skip


@pytest.mark.gpu
def test_destripe_zarr_end_to_end(tmp_path):
    """Whole chunk map on a 20-plane synthetic tile, two 'ranks' run one after the other."""
    from oracle import destripe_oracle as orc

    stack = synth.synthetic_stack(20, 96, 128, n_unique=5)
    src = MiniZarrArray.create(str(tmp_path / "X_0_Y_0.zarr"), (1, 1) + stack.shape, (1, 1, 8, 32, 32), np.uint16,
                               compressor="zlib")  # fmt: skip
    src[0, 0] = stack
    total = 0
    for rank in range(2):
        n, _ = zd.destripe_zarr_store(str(tmp_path / "X_0_Y_0.zarr"), str(tmp_path / "out.zarr"), synth.CELLS_CONFIG,
                                synth.NO_CELLS_CONFIG, prediction_chunksize=(8, 96, 128),
                                output_chunks=(1, 1, 8, 32, 32), rank=rank, world_size=2, device=0)  # fmt: skip
        total += n
    assert total == 20
    out = MiniZarrArray.open(str(tmp_path / "out.zarr"))[0, 0]
    assert out.dtype == np.uint16 and out.shape == stack.shape
    for z in range(20):
        ref = orc.filter_stripes(stack[z], "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, 2500)
        d = np.abs(out[z].astype(np.int64) - np.clip(ref, 0, 65535).astype(np.uint16).astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (z, int(d.max()))


@pytest.mark.gpu
def test_chunk_map_at_production_geometry_against_the_oracle(tmp_path):
    """BASELINE configs[3] at the geometry production runs it in (zarr_destriper.py:253-336, 1252-1267 of the reference):
    the 1600 x 2000 tile, 64-plane blocks (``prediction_chunksize=(64, H, W)``), chunks ``(1, 1, 64, 128, 128)``,
    Blosc-zstd in and out, dark / flat-field correction on.  192 planes = 3 blocks: both staging buffer sets wrap, every
    block is split into 4 stream parts INSIDE the 3-stream upload / compute / download pipeline, with the deferred joins
    between ``run_device`` and ``planes_to_bricks``.

    * one plane of EVERY stream part of EVERY block under the parity statement proper (``parity_util.u16_plane_parity``:
      the plane's float32 result satisfies ``check_plane`` -- every pixel within 1e-4 of the oracle with the counted
      near-threshold decisions forced -- and EVERY stored pixel is that result through the reference's
      ``flatfield_correction`` arithmetic within one count; no outlier allowance),
    * planes 0 and 1 against the samples the real reference wrote (tests/golden/large_stats.npz; pushed through
      flatfield_correction's arithmetic),
    * the whole store byte-identical to the host gather / scatter path (execute_worker -> destripe_planes)."""
    from parity_util import stream_part_picks, u16_plane_parity

    H, W, Z, BZ = 1600, 2000, 192, 64
    bank = synth.synthetic_bank(8, H, W)
    vol = synth.synthetic_stack(Z, H, W, bank=bank)
    yy, xx = np.mgrid[0:H, 0:W]
    r2 = ((yy - H / 2.0) / (H / 2.0)) ** 2 + ((xx - W / 2.0) / (W / 2.0)) ** 2
    flat = (1.0 - 0.15 * r2).astype(np.float32)
    dark = np.full((H, W), 100.0, dtype=np.float32)
    sc = {"retrospective": True, "flatfield": flat, "darkfield": dark}
    name = "X_0_Y_0.zarr"
    src = MiniZarrArray.create(str(tmp_path / name), (1, 1, Z, H, W), (1, 1, 64, 128, 128), np.uint16, compressor="blosc")
    for z in range(0, Z, BZ):
        src[0, 0, z : z + BZ] = vol[z : z + BZ]
    outs = {}
    for mode in (True, False):
        path = str(tmp_path / "out_{}.zarr".format(int(mode)))
        n, _ = zd.destripe_zarr_store(str(tmp_path / name), path, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, sc,
                                prediction_chunksize=(BZ, H, W), output_chunks=(1, 1, 64, 128, 128), device=0,
                                device_retile=mode, compressor="blosc", io_threads=16)  # fmt: skip
        assert n == Z
        outs[mode] = MiniZarrArray.open(path)
        assert outs[mode].compressor[0] == "blosc"
    zd.release_staging()
    dev = outs[True][0, 0]
    assert dev.shape == (Z, H, W) and dev.dtype == np.uint16
    # (1) every stream part of every block
    picks = [z for b in range(0, Z, BZ) for z in stream_part_picks(BZ, b, b + BZ)]
    assert len(picks) == 12
    from aind_smartspim_destripe_amd import engine as eng_mod

    e = eng_mod.DestripeEngine(0)
    try:
        for z in picks:  # the parity statement proper: float32 result under check_plane, stored value within one count of it
            st = u16_plane_parity(e, dev[z], vol[z], name.replace(".zarr", ""), sc, ("chunk map", z))
            print("[chunk map] plane {:3d}: flips per level {}, {} px beyond 1e-4 unforced, off by one count {:.2e}".format(
                z, st["flips"], st["beyond_unforced"], st["off_by_one"]))
    finally:
        e.close()
    # (2) the real reference's samples of bank planes 0, 1 (= stack planes 0, 1)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_stats.npz"), allow_pickle=False)
    rs = np.random.RandomState(7)
    sy, sx = rs.randint(0, H, 4096), rs.randint(0, W, 4096)
    for k in (0, 1):
        ref = g["s1600x2000__k{}__u16__sample".format(k)]
        d_ = dark[sy, sx].astype(np.float64)
        ref = np.where(ref > d_, ref - d_, 0.0) / flat[sy, sx].astype(np.float64)
        want = np.clip(ref, 0, 65535).astype(np.uint16).astype(np.int64)
        diff = np.abs(dev[k][sy, sx].astype(np.int64) - want)
        assert int((diff > np.maximum(1, 2e-4 * want)).sum()) <= 4, (k, int(diff.max()))  # k1 has one known flip
    # (3) device brick path == host gather / scatter path, byte for byte
    np.testing.assert_array_equal(dev, outs[False][0, 0])
