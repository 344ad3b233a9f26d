"""Rows f2 / f4 of SURVEY section 8: shading plumbing and the TIFF / RAW directory mode.

CPU tests: the TIFF codec against files written by the real ``tifffile`` (``oracle/make_golden_tiff.py``), the
readers with the reference's own test cases (``code/tests/test_readers.py``, ``test_zarr_destriper.py:25-47``),
flat discovery, and ``batch_filter`` with the GPU call patched (as the reference patches the filter in
``test_filtering.py:242-281``).  GPU tests (``-m gpu``): the stand-alone ``flatfield_correction`` against the
golden vectors and the directory / channel drivers end to end against the CPU oracle.
"""

import json
import os
from pathlib import Path
from unittest.mock import mock_open, patch

import numpy as np
import pytest

from aind_smartspim_destripe_amd import destriper, mini_tiff, readers, synth
from aind_smartspim_destripe_amd import zarr_destriper as zd
from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

TIFF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiff")
CASES = {  # file -> (seed, shape, dtype) of oracle/make_golden_tiff.py
    "u16_le.tif": (1, (37, 53), np.uint16),
    "u16_be.tif": (2, (37, 53), np.uint16),
    "u16_deflate.tif": (3, (37, 53), np.uint16),
    "u16_deflate_pred.tif": (4, (37, 53), np.uint16),
    "u16_big.tif": (5, (37, 53), np.uint16),
    "u16_pages.tif": (6, (3, 20, 24), np.uint16),
    "f32.tif": (7, (37, 53), np.float32),
    "u16_tiled.tif": (8, (37, 53), np.uint16),
    "u8_strips.tif": (9, (37, 53), np.uint8),
}


def _plane(seed, shape, dtype):
    rs = np.random.RandomState(seed)
    if np.dtype(dtype).kind == "f":
        return rs.rand(*shape).astype(dtype)
    return rs.randint(0, np.iinfo(dtype).max + 1, shape).astype(dtype)


# ------------------------------------------------------------------------------------------ TIFF


@pytest.mark.parametrize("name", sorted(CASES))
def test_mini_tiff_reads_tifffile_output(name, tmp_path):
    seed, shape, dtype = CASES[name]
    want = _plane(seed, shape, dtype)
    got = mini_tiff.imread(os.path.join(TIFF_DIR, name))
    assert got.dtype == want.dtype and got.dtype.isnative
    np.testing.assert_array_equal(got, want)
    for comp in (None, 1):  # and our own writer round-trips
        mini_tiff.imwrite(str(tmp_path / "rt.tiff"), want, compression=comp)
        back = mini_tiff.imread(str(tmp_path / "rt.tiff"))
        assert back.dtype == want.dtype
        np.testing.assert_array_equal(back, want)


def test_mini_tiff_errors(tmp_path):
    p = tmp_path / "bad.tif"
    p.write_bytes(b"not a tiff at all")
    with pytest.raises(mini_tiff.TiffError):
        mini_tiff.imread(str(p))
    with pytest.raises(FileNotFoundError):
        mini_tiff.imread(str(tmp_path / "missing.tif"))
    with pytest.raises(ValueError):
        mini_tiff.imwrite(str(p), np.zeros((2, 2, 2, 2), np.uint16))
    with pytest.raises(ValueError):
        mini_tiff.imwrite(str(p), np.zeros((2, 2), np.complex64))
    # a truncated file must not read out of bounds
    raw = open(os.path.join(TIFF_DIR, "u16_le.tif"), "rb").read()
    p.write_bytes(raw[:2000] + raw[-300:])
    with pytest.raises(Exception):
        mini_tiff.imread(str(p))


def test_written_tiff_is_readable_by_tifffile(tmp_path):
    """Cross-check with the real library when the reference interpreter is around (this container)."""
    import subprocess

    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        pytest.skip("no interpreter with tifffile")
    a = _plane(21, (64, 48), np.uint16)
    mini_tiff.imwrite(str(tmp_path / "a.tiff"), a)
    mini_tiff.imwrite(str(tmp_path / "s.tiff"), np.stack([a, a + 1]), compression=6)
    np.save(str(tmp_path / "a.npy"), a)
    code = (
        "import sys, numpy as np, tifffile; d = sys.argv[1]; a = np.load(d + '/a.npy');"
        "assert np.array_equal(tifffile.imread(d + '/a.tiff'), a);"
        "s = tifffile.imread(d + '/s.tiff'); assert s.shape == (2, 64, 48) and np.array_equal(s[1], a + 1)"
    )
    r = subprocess.run([py, "-c", code, str(tmp_path)], capture_output=True, text=True, timeout=120)
    if "No module named" in r.stderr:
        pytest.skip("tifffile not importable there")
    assert r.returncode == 0, r.stderr


# --------------------------------------------------------------------------------------- readers


def test_get_extension():  # code/tests/test_readers.py:22-30
    assert readers._get_extension("image.tif") == ".tif"
    assert readers._get_extension("/path/to/image.png") == ".png"
    assert readers._get_extension("C:\\Images\\image.raw") == ".raw"
    assert readers._get_extension("no_extension") == ""


def test_raw_imread_endianness(tmp_path):
    """The cases of code/tests/test_readers.py:32-66 on real files: a 300 x 200 plane whose header reads as a
    smaller width in its own byte order than in the other one; a missing file raises."""
    for order in (">", "<"):
        p = tmp_path / "p.raw"
        with open(p, "wb") as f:
            f.write(np.array([300, 200], dtype=order + "u4").tobytes())
            f.write(np.ones((300, 200), dtype=order + "u2").tobytes())
        r = readers.raw_imread(str(p))
        assert r.shape == (300, 200) and r.dtype == np.dtype(order + "u2") and int(r.max()) == 1
    assert readers._raw_geometry(np.array([100, 50], dtype="<u4").tobytes()) == (100, 50, np.dtype("<u2"))
    with pytest.raises(OSError):
        readers.raw_imread(str(tmp_path / "invalid_path.raw"))
    (tmp_path / "short.raw").write_bytes(b"abc")
    with pytest.raises(ValueError):
        readers.raw_imread(str(tmp_path / "short.raw"))


def test_imread_dispatch_and_real_files(tmp_path):
    with patch("aind_smartspim_destripe_amd.readers.raw_imread", return_value=np.zeros((10, 10))), patch(
        "aind_smartspim_destripe_amd.mini_tiff.imread", return_value=np.ones((10, 10))
    ):  # the dispatch of code/tests/test_readers.py:68-80
        assert np.array_equal(readers.imread("image.raw"), np.zeros((10, 10)))
        assert np.array_equal(readers.imread("image.tif"), np.ones((10, 10)))
        assert np.array_equal(readers.imread(Path("image.tiff")), np.ones((10, 10)))
    assert readers.imread("image.jpg") is None
    with patch("aind_smartspim_destripe_amd.mini_png.imread", return_value=np.full((10, 10), 2)):
        assert np.array_equal(readers.imread("image.png"), np.full((10, 10), 2))
    # a real little-endian and a real big-endian .raw plane
    a = _plane(3, (40, 40), np.uint16)
    for order in ("<", ">"):
        p = tmp_path / "p{}.raw".format("le" if order == "<" else "be")
        with open(p, "wb") as f:
            f.write(np.array([40, 40], dtype=order + "u4").tobytes())
            f.write(a.astype(order + "u2").tobytes())
        np.testing.assert_array_equal(np.asarray(readers.imread(str(p))), a)


# ------------------------------------------------------------------------------ shading plumbing


def test_read_json_as_dict():  # code/tests/test_zarr_destriper.py:25-47
    with patch("builtins.open", mock_open(read_data='{"key": "value"}')), patch("os.path.exists", return_value=True):
        assert zd.read_json_as_dict("fake_path.json") == {"key": "value"}
    with patch("builtins.open", side_effect=UnicodeDecodeError("utf-8", b"", 0, 1, "error")), patch(
        "os.path.exists", return_value=True
    ):
        with pytest.raises(UnicodeDecodeError):
            zd.read_json_as_dict("fake_path.json")
    with patch("os.path.exists", return_value=False):
        assert zd.read_json_as_dict("fake_path.json") == {}


def _derivatives(tmp_path, n_flats=2, tile_config=True, side_missing=False):
    d = tmp_path / "derivatives"
    d.mkdir()
    cfg = {
        "t0": {"Laser": "488", "X": "431040", "Y": "368180", "Side": "0"},
        "t1": {"Laser": "488", "X": "431040", "Y": "394100", "Side": "1"},
        "t2": {"Laser": "561", "X": "431040", "Y": "368180", "Side": "1"},
        "t3": {"Laser": "488", "X": "465600", "Y": "368180", "Side": "1"},
    }
    if side_missing:
        del cfg["t1"]["Side"]
    meta = {"tile_config": cfg} if tile_config else {"other": 1}
    (d / "metadata.json").write_text(json.dumps(meta))
    flats = []
    for i, name in enumerate(["FlatReal488_10.tif", "FlatReal488_9.tif", "FlatReal488_11.tif"][:n_flats]):
        f = (1000 + 100 * i + np.arange(24 * 32).reshape(24, 32)).astype(np.uint16)
        mini_tiff.imwrite(str(d / name), f)
        flats.append((name, f))
    mini_tiff.imwrite(str(d / "FlatReal561_0.tif"), np.zeros((24, 32), np.uint16))
    mini_tiff.imwrite(str(d / "DarkMaster_cropped.tif"), np.full((30, 40), 100, np.uint16))
    return d, dict(flats)


def test_get_microscope_flats(tmp_path):
    d, flats = _derivatives(tmp_path)
    got, cfg = zd.get_microscope_flats("Ex_488_Em_525", d)
    assert cfg == {"431040": {"368180": 0, "394100": 1}, "465600": {"368180": 1}}
    # natural order: FlatReal488_9 before FlatReal488_10
    np.testing.assert_array_equal(got[0], flats["FlatReal488_9.tif"])
    np.testing.assert_array_equal(got[1], flats["FlatReal488_10.tif"])
    assert zd.get_microscope_flats("NoWavelength", d) == (None, None)
    assert zd.get_microscope_flats("Ex_488_Em_525", tmp_path / "nowhere") == (None, None)


def test_get_microscope_flats_errors(tmp_path):
    (tmp_path / "a").mkdir(), (tmp_path / "b").mkdir(), (tmp_path / "c").mkdir()
    d, _ = _derivatives(tmp_path / "a", n_flats=3)
    with pytest.raises(ValueError):
        zd.get_microscope_flats("Ex_488_Em_525", d)
    d, _ = _derivatives(tmp_path / "b", tile_config=False)
    with pytest.raises(ValueError):
        zd.get_microscope_flats("Ex_488_Em_525", d)
    d, _ = _derivatives(tmp_path / "c", side_missing=True)
    with pytest.raises(KeyError):
        zd.get_microscope_flats("Ex_488_Em_525", d)


def test_load_shadow_correction(tmp_path):
    d, flats = _derivatives(tmp_path)
    out = tmp_path / "results" / "Ex_488_Em_525" / "431040_368180.zarr"
    sc = zd.load_shadow_correction(d, out)  # microscope flats, normalised into [1, 2] (float16)
    assert sc["retrospective"] is False and sc["tile_config"]["431040"]["394100"] == 1
    assert sc["flatfield"].shape == (2, 24, 32) and sc["flatfield"].dtype == np.float16
    assert float(sc["flatfield"].min()) == 1.0 and float(sc["flatfield"].max()) == 2.0
    assert sc["darkfield"].shape == (30, 40) and int(sc["darkfield"][0, 0]) == 100
    retro = np.ones((24, 32), np.float32)
    sc = zd.load_shadow_correction(d, out, flatfield=retro)
    assert sc["retrospective"] is True and sc["flatfield"] is retro and sc["tile_config"] is None
    os.remove(d / "DarkMaster_cropped.tif")
    with pytest.raises(FileNotFoundError):
        zd.load_shadow_correction(d, out, flatfield=retro)
    sc = zd.load_shadow_correction(tmp_path / "nowhere", out, flatfield=retro)
    assert sc["darkfield"] is None and sc["retrospective"] is True


# --------------------------------------------------------------------------------- directory mode


def test_imsave_naming_and_errors(tmp_path):
    a = _plane(5, (16, 20), np.uint16)
    destriper.imsave(str(tmp_path / "a.tif"), a)
    destriper.imsave(str(tmp_path / "b.raw"), a)
    destriper.imsave(str(tmp_path / "c.raw"), a, output_format=".tif")
    assert sorted(os.listdir(tmp_path)) == ["a.tiff", "b.tiff", "c.tif"]
    np.testing.assert_array_equal(mini_tiff.imread(str(tmp_path / "b.tiff")), a)
    with pytest.raises(NotImplementedError):
        destriper.imsave(str(tmp_path / "d.jpg"), a)
    with pytest.raises(ValueError):
        destriper.imsave(str(tmp_path / "d.tif"), a, output_format=".jpg")


def _png_plane(seed, shape, dtype):  # the arrays of oracle/make_golden_png.py
    rs = np.random.RandomState(seed)
    base = np.add.outer(np.arange(shape[0]) * 7, np.arange(shape[1]) * 3)
    if len(shape) == 3:
        base = base[..., None] + np.arange(shape[2]) * 11
    hi = np.iinfo(dtype).max
    return ((base * (hi // 512) + rs.randint(0, hi // 64 + 2, shape)) % (hi + 1)).astype(dtype)


PNG_CASES = {
    "u16_gray.png": (1, (37, 53), np.uint16),
    "u16_gray_c9.png": (2, (64, 40), np.uint16),
    "u8_gray.png": (3, (29, 31), np.uint8),
    "u8_rgb.png": (4, (20, 24, 3), np.uint8),
    "u8_rgba.png": (5, (18, 22, 4), np.uint8),
    "u16_gray_c0.png": (6, (33, 17), np.uint16),
}


def test_png_interlaced_palette_and_converted_layouts_as_the_real_imageio_reads_them():
    """The PNG corners the reference gets from its library (``iio.imread``, ``readers.py:86-87``; VERDICT r3 missing #4):
    Adam7-interlaced files of every layout, palette images (1 ... 8 bits, ``tRNS``, grey and colour palettes, plain and
    interlaced) and the layouts Pillow converts on the way in (1 / 2 / 4-bit grey -> 8 bits, grey + alpha -> RGBA, 16-bit
    colour -> high bytes).  imageio / Pillow cannot WRITE most of these, so ``oracle/make_golden_png.py`` encodes them by
    hand, reads them with the REAL imageio 2.9.0 and stores what it returns (``expected_r4.npz``): ``mini_png.imread`` must
    return the same array, shape and dtype, for all 25 files."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png")
    exp = np.load(os.path.join(root, "expected_r4.npz"), allow_pickle=False)
    assert len(exp.files) == 25
    seen = {"interlaced": 0, "palette": 0, "rgba": 0, "grey2d": 0}
    for name in exp.files:
        want = exp[name]
        got = readers.imread(os.path.join(root, name))
        assert got.dtype == want.dtype and got.shape == want.shape, (name, got.dtype, got.shape, want.dtype, want.shape)
        np.testing.assert_array_equal(got, want, err_msg=name)
        seen["interlaced"] += name.startswith("a7_")
        seen["palette"] += "p" in name.split("_")[0] or "_p" in name
        seen["rgba"] += want.ndim == 3 and want.shape[2] == 4
        seen["grey2d"] += want.ndim == 2
    assert seen["interlaced"] >= 9 and seen["palette"] >= 11 and seen["rgba"] >= 10 and seen["grey2d"] >= 9, seen
    # a truncated interlaced file and an unknown interlace method are refused, not mis-read
    raw = open(os.path.join(root, "a7_u8_rgb.png"), "rb").read()
    import struct
    import zlib

    i = raw.index(b"IHDR")
    bad = bytearray(raw)
    bad[i + 4 + 12] = 2  # interlace method byte
    bad[i + 4 + 13 : i + 4 + 17] = struct.pack(">I", zlib.crc32(bytes(bad[i : i + 4 + 13])) & 0xFFFFFFFF)
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "bad.png")
        open(path, "wb").write(bytes(bad))
        with pytest.raises(ValueError, match="interlace"):
            readers.imread(path)


def test_png_reader_against_files_of_the_real_imageio_and_writer_against_pillow(tmp_path):
    """PNG planes (reference: iio.imread, readers.py:86-87; iio.v3.imwrite(..., compress_level=), destriper.py:107-110).
    tests/golden/png/* were written by the real imageio 2.9.0 (oracle/make_golden_png.py); what mini_png writes is read
    back by itself and by the Pillow of this image."""
    from aind_smartspim_destripe_amd import mini_png

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png")
    filters_seen = set()
    for name, (seed, shape, dtype) in PNG_CASES.items():
        want = _png_plane(seed, shape, dtype)
        got = readers.imread(os.path.join(root, name))
        assert got.dtype == want.dtype and got.shape == want.shape, name
        np.testing.assert_array_equal(got, want, err_msg=name)
        for level in (0, 1, 9):
            out = str(tmp_path / "w{}_{}".format(level, name))
            mini_png.imwrite(out, want, compress_level=level)
            np.testing.assert_array_equal(mini_png.imread(out), want)
            try:
                from PIL import Image
            except ImportError:  # pragma: no cover
                continue
            with Image.open(out) as im:
                pil = np.array(im)
            np.testing.assert_array_equal(pil.astype(want.dtype), want, err_msg="Pillow reads " + name)
        import zlib as _z

        raw = open(os.path.join(root, name), "rb").read()
        i = raw.index(b"IDAT")
        n = int.from_bytes(raw[i - 4 : i], "big")
        rows = _z.decompressobj().decompress(raw[i + 4 : i + 4 + n])
        stride = len(rows) // shape[0]
        filters_seen |= {rows[k * stride] for k in range(shape[0])}
    assert filters_seen >= {1, 2, 4}, filters_seen  # the real files exercise the sequential filters of the reader
    # directory-mode API: imsave with output_format=".png" (destriper.py:107-110), read back through imread
    a = _png_plane(7, (16, 20), np.uint16)
    destriper.imsave(str(tmp_path / "x.tif"), a, compression=3, output_format=".png")
    np.testing.assert_array_equal(readers.imread(str(tmp_path / "x.png")), a)
    destriper.imsave(str(tmp_path / "y.png"), a)  # a PNG source is saved as <stem>.tiff without output_format
    np.testing.assert_array_equal(mini_tiff.imread(str(tmp_path / "y.tiff")), a)
    with pytest.raises(ValueError, match="not a PNG"):
        open(tmp_path / "bad.png", "wb").write(b"nope")
        readers.imread(str(tmp_path / "bad.png"))
    with pytest.raises(TypeError):
        mini_png.imwrite(str(tmp_path / "f.png"), a.astype(np.float32))


def _image_tree(root, n=5, shape=(64, 96)):
    (root / "X_0" / "X_0_Y_0").mkdir(parents=True)
    (root / "X_0" / "X_0_Y_1").mkdir(parents=True)
    planes = {}
    for k in range(n):
        sub = "X_0_Y_0" if k % 2 == 0 else "X_0_Y_1"
        p = root / "X_0" / sub / "{:06d}.tif".format(k)
        a = synth.synthetic_plane(k, *shape)
        mini_tiff.imwrite(str(p), a)
        planes[p.relative_to(root)] = a
    (root / "X_0" / "X_0_Y_0" / "broken.tif").write_bytes(b"II*\0garbage")
    (root / "notes.txt").write_text("acquisition notes")
    (root / "ASI.ini").write_text("[stage]")
    (root / "ignored.dat").write_text("x")
    return planes


def test_find_all_images_and_batch_filter_patched(tmp_path):
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir(), dst.mkdir()
    planes = _image_tree(src)
    found = destriper._find_all_images(src, src, dst)
    assert len(found) == 6 and (dst / "X_0" / "X_0_Y_1").is_dir()
    calls = []

    def fake(stack, **kw):
        calls.append((stack.shape, kw["input_tile_path"], kw["out_dtype"], kw.get("microscope_high_int", 2700)))
        return (stack // 2).astype(kw["out_dtype"])

    with patch("aind_smartspim_destripe_amd.filtering.destripe_planes", side_effect=fake):
        n = destriper.batch_filter(src, dst, workers=2, chunks=4, high_int_filt_params=synth.CELLS_CONFIG,
                                   low_int_filt_params=synth.NO_CELLS_CONFIG, shadow_correction=None)  # fmt: skip
    assert n == 5 and sum(c[0][0] for c in calls) == 5
    assert all(c[2] == np.uint16 and c[3] == 2700 for c in calls)  # uint16 sources, default high_int of the TIFF path
    for rel, a in planes.items():
        got = mini_tiff.imread(str((dst / rel).with_suffix(".tiff")))
        assert got.dtype == np.uint16
        np.testing.assert_array_equal(got, a // 2)
    log = (dst / "destripe_log.txt").read_text()
    assert "broken.tif" in log and log.startswith("Error reading the following images")
    assert (dst / "notes.txt").read_text() == "acquisition notes" and (dst / "ASI.ini").exists()
    assert not (dst / "ignored.dat").exists()


# --------------------------------------------------------------------------------------------- GPU


@pytest.mark.gpu
def test_gpu_flatfield_correction_per_row_baseline():
    """``baseline[:, np.newaxis]`` (filtering.py:393-398, 409): a 2-D plane takes one baseline value per ROW, a
    [1, H, W] stack one per plane; a baseline of another length fails to broadcast (ValueError, as NumPy's)."""
    from aind_smartspim_destripe_amd import filtering
    from oracle import destripe_oracle as orc

    rs = np.random.RandomState(5)
    for dtype in (np.uint16, np.float32):
        img = rs.randint(0, 4000, (37, 52)).astype(dtype)
        flat = (0.7 + 0.6 * rs.rand(37, 52)).astype(np.float32)
        dark = (90 + 20 * rs.rand(40, 60)).astype(np.float32)  # larger than the plane: cropped (:377)
        base = (50.0 * rs.rand(37)).astype(np.float64)
        out = filtering.flatfield_correction(img.copy(), flat, dark, base)
        ref = orc.flatfield_correction(img.copy(), flat, dark, base)
        assert out.dtype == np.uint16 and out.shape == ref.shape
        d = np.abs(out.astype(np.int64) - ref.astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 5e-3, (dtype, int(d.max()))
        assert np.abs(ref.astype(np.int64) - orc.flatfield_correction(img.copy(), flat, dark).astype(np.int64)).max() > 10
        one = filtering.flatfield_correction(img[None].copy(), flat[None], dark[:37, :52][None], np.array([7.0]))
        ref1 = orc.flatfield_correction(img[None].copy(), flat[None], dark[:37, :52][None], np.array([7.0]))
        assert np.abs(one.astype(np.int64) - ref1.astype(np.int64)).max() <= 1
        with pytest.raises(ValueError):
            filtering.flatfield_correction(img.copy(), flat, dark, np.zeros(5))


@pytest.mark.gpu
def test_gpu_flatfield_correction_golden(golden_misc):
    from aind_smartspim_destripe_amd import filtering

    g = golden_misc
    out = filtering.flatfield_correction(
        np.array([[[10, 20], [30, 40]]]), np.array([[[2, 2], [2, 2]]]), np.array([[[1, 1], [1, 1]]])
    )  # code/tests/test_filtering.py:226-240
    assert out.dtype == np.uint16
    np.testing.assert_array_equal(out, g["flat_kat"])
    with pytest.raises(ValueError):
        filtering.flatfield_correction(
            np.array([[[10, 20], [30, 40]]]), np.array([[[2, 2], [2, 2]]]), np.array([[[1, 1]]])
        )
    with pytest.raises(ValueError):
        filtering.flatfield_correction(np.zeros((4, 4)), np.ones((3, 4)), np.zeros((4, 4)))
    out = filtering.flatfield_correction(g["flat_f__img"], g["flat_f__flat"], g["flat_f__dark"])
    d = np.abs(out.astype(np.int64) - g["flat_f__out"].astype(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3  # float32 division next to a truncation boundary
    # integer planes truncate after the dark subtraction (filtering.py:400-403)
    img = np.array([[1000, 50], [300, 65535]], dtype=np.uint16)
    dark = np.array([[100.5, 60.0], [0.25, 0.0]])
    flat = np.array([[1.0, 1.0], [0.5, 0.5]])
    np.testing.assert_array_equal(filtering.flatfield_correction(img, flat, dark), [[899, 0], [598, 65535]])


@pytest.mark.gpu
def test_gpu_batch_filter_and_read_filter_save(tmp_path):
    from oracle import destripe_oracle as orc

    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir(), dst.mkdir()
    planes = _image_tree(src, n=5, shape=(96, 128))
    n = destriper.batch_filter(src, dst, workers=4, chunks=3, high_int_filt_params=synth.CELLS_CONFIG,
                               low_int_filt_params=synth.NO_CELLS_CONFIG, shadow_correction=None)  # fmt: skip
    assert n == 5

    def check(path, a):
        got = mini_tiff.imread(str(path))
        ref = orc.filter_stripes(a, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None)  # default 2700
        d = np.abs(got.astype(np.int64) - ref.astype(np.uint16).astype(np.int64))
        assert got.dtype == np.uint16 and d.max() <= 1 and (d > 0).mean() < 2e-3

    for rel, a in planes.items():
        check((dst / rel).with_suffix(".tiff"), a)
    rel, a = next(iter(planes.items()))
    destriper.read_filter_save(str(dst), str(src / rel), str(dst / "single.tif"), synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG)
    check(dst / "single.tiff", a)
    # an unreadable file is logged and skipped
    destriper.read_filter_save(str(dst), str(src / "X_0" / "X_0_Y_0" / "broken.tif"), str(dst / "b.tif"),
                               synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG)  # fmt: skip
    assert not (dst / "b.tiff").exists()
    # PNG planes in, PNG planes out (readers.py:86-87, destriper.py:107-110)
    from aind_smartspim_destripe_amd import mini_png

    psrc, pdst = tmp_path / "pin", tmp_path / "pout"
    (psrc / "X_1").mkdir(parents=True), pdst.mkdir()
    pngs = {}
    for k in (1, 2):
        pngs[k] = synth.synthetic_plane(k, 96, 128)
        mini_png.imwrite(str(psrc / "X_1" / "{:06d}.png".format(k)), pngs[k], compress_level=1)
    n = destriper.batch_filter(psrc, pdst, workers=2, chunks=2, high_int_filt_params=synth.CELLS_CONFIG,
                               low_int_filt_params=synth.NO_CELLS_CONFIG, shadow_correction=None, output_format=".png")  # fmt: skip
    assert n == 2
    for k, a in pngs.items():
        got = mini_png.imread(str(pdst / "X_1" / "{:06d}.png".format(k)))
        ref = orc.filter_stripes(a, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None)
        d = np.abs(got.astype(np.int64) - ref.astype(np.uint16).astype(np.int64))
        assert got.dtype == np.uint16 and d.max() <= 1 and (d > 0).mean() < 2e-3


@pytest.mark.gpu
def test_gpu_destripe_channel_with_shading_and_pyramid(tmp_path):
    """Two tiles of one channel, retrospective flat per laser side, microscope dark, pyramid levels."""
    from oracle import destripe_oracle as orc
    from oracle import format_oracle as fo

    H, W, Z = 64, 96, 12
    chan = tmp_path / "data" / "Ex_488_Em_525"
    tiles = {"431040_368180": 0, "431040_394100": 1}
    stacks = {}
    for t, (name, side) in enumerate(tiles.items()):
        stack = synth.synthetic_stack(Z, H, W, n_unique=4) + np.uint16(t)
        a = MiniZarrArray.create(str(chan / (name + ".zarr") / "0"), (1, 1, Z, H, W), (1, 1, 4, 32, 32), np.uint16,
                                 compressor="zlib")  # fmt: skip
        a[0, 0] = stack
        stacks[name] = stack
    d = tmp_path / "derivatives"
    d.mkdir()
    mini_tiff.imwrite(str(d / "DarkMaster_cropped.tif"), np.full((H + 8, W + 8), 90, np.uint16))
    yy, xx = np.mgrid[0:H, 0:W]
    flats = []
    for side in (0, 1):
        f = (1.0 + 0.2 * side - 0.3 * ((yy - H / 2) / H) ** 2 - 0.2 * ((xx - W / 2) / W) ** 2).astype(np.float32)
        mini_tiff.imwrite(str(d / "flat_{}.tif".format(side)), f)
        flats.append(f)
    params = {"cells_config": synth.CELLS_CONFIG, "no_cells_config": synth.NO_CELLS_CONFIG}
    with pytest.raises(ValueError):
        zd.destripe_channel(tmp_path / "data", d, "Ex_488_Em_525", tmp_path / "results", [1.8, 1.8, 2.0],
                            [d / "flat_0.tif", d / "flat_1.tif"], {"0": ["431040_368180"]}, params,
                            prediction_chunksize=(4, H, W))  # fmt: skip
    # the reference's caller, keyword for keyword (run_capsule.py:394-403), plus the engine's keyword-only extras
    done = zd.destripe_channel(zarr_dataset_path=tmp_path / "data", channel_name="Ex_488_Em_525",
                               results_folder=tmp_path / "results2", derivatives_path=d, xyz_resolution=[1.8, 1.8, 2.0],
                               estimated_channel_flats=[d / "flat_0.tif", d / "flat_1.tif"],
                               laser_tiles={"0": ["431040_368180"], "1": ["431040_394100"]}, parameters=params,
                               prediction_chunksize=(4, H, W), output_chunks=(1, 1, 4, 32, 32),
                               compressor="zlib")  # fmt: skip
    assert done == {"431040_368180.zarr": Z, "431040_394100.zarr": Z}
    dark = np.full((H + 8, W + 8), 90, np.uint16)
    for name, side in tiles.items():
        out_dir = tmp_path / "results2" / "destriped_data" / "Ex_488_Em_525" / (name + ".zarr")
        got = MiniZarrArray.open(str(out_dir / "0"))[0, 0]
        sc = {"retrospective": True, "flatfield": flats[side], "darkfield": dark, "tile_config": None}
        for z in range(Z):
            ref = orc.filter_stripes(stacks[name][z], name, synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, sc, 2500)
            dd = np.abs(got[z].astype(np.int64) - ref.astype(np.int64))
            assert dd.max() <= 1 and (dd > 0).mean() < 2e-3, (name, z, int(dd.max()))
        pyr = fo.pyramid(got, 3)
        for lvl in (1, 2):
            np.testing.assert_array_equal(MiniZarrArray.open(str(out_dir / str(lvl)))[0, 0], pyr[lvl])


@pytest.mark.gpu
def test_gpu_foreground_background_mean_golden(golden_misc):
    """Stand-alone a2 (``filtering.py:54-88``) on the device against vectors from the real reference."""
    from aind_smartspim_destripe_amd import filtering

    g = golden_misc
    allv = np.arange(65536, dtype=np.uint16)
    for dt, x in (("u16", allv), ("f32", allv.astype(np.float32))):
        fore, back, mask = filtering.get_foreground_background_mean(x)
        np.testing.assert_allclose([fore, back], g["fgbg_all__{}__means".format(dt)], rtol=1e-6)
        assert mask.dtype == np.float16 and mask.shape == x.shape
        np.testing.assert_array_equal(mask.astype(np.uint8), np.unpackbits(g["fgbg_all__{}__mask".format(dt)])[:65536])
    fr = g["fgbg_frac__in"]  # fractional float32 pixels around the cut-off
    fore, back, mask = filtering.get_foreground_background_mean(fr)
    np.testing.assert_array_equal(mask.astype(np.uint8), g["fgbg_frac__mask"])
    np.testing.assert_allclose([fore, back], g["fgbg_frac__means"], rtol=1e-6)
    # the host-side cutoff reproduces the decision for every float16 bit pattern at the default threshold
    h = np.arange(65536, dtype=np.uint16).view(np.float16)
    table = np.unpackbits(g["f16_mask_table"])[:65536].astype(bool)
    ok = np.isfinite(h)
    np.testing.assert_array_equal((h >= np.float16(filtering._foreground_cutoff(0.3)))[ok], table[ok])
    # empty / all-background / all-foreground (code/tests/test_filtering.py:68-114)
    f, b, m = filtering.get_foreground_background_mean(np.array([]))
    assert f == 0.0 and b == 0.0 and m.size == 0
    f, b, m = filtering.get_foreground_background_mean(np.array([10, 20, 30, 40, 50]), 1.0)
    assert f == 0.0 and b == 30.0 and not m.any()
    f, b, m = filtering.get_foreground_background_mean(np.array([400, 420, 430, 440, 460]), 0.0)
    assert f == 430.0 and b == 0.0 and m.all()
    img = np.array([[100, 200], [500, 700]], dtype=np.uint16)
    f, b, m = filtering.get_foreground_background_mean(img)
    assert (f, b) == (600.0, 150.0) and m.tolist() == [[0.0, 0.0], [1.0, 1.0]]


@pytest.mark.gpu
def test_gpu_two_ranks_destripe_channel_with_shading_on_the_one_gpu(tmp_path):
    """Rehearsal of the multi-rank chunk map with shading: two processes -- one per rank, both on this
    box's one GPU -- run ``destripe_channel`` with the reference's keyword set and a ``RankGroup``.  RCCL refuses two ranks
    on one device, so the ranks agree on the host transport and the planes travel through the rendezvous directory; on
    a real node the same calls are RCCL broadcasts.  Rank 0 alone reads the flats and the dark plane, rank 1 corrects its
    z-range with the copies it was sent; the store must equal the oracle's correction and a one-rank run; the pyramid is
    written once, after both ranks.  A third run with ONE rank and ``DSX_FORCE_COMM=1`` sends the same planes through a
    real RCCL communicator (``ncclBroadcast`` on the device)."""
    import json
    import subprocess
    import sys

    from oracle import destripe_oracle as orc
    from oracle import format_oracle as fo

    H, W, Z = 64, 96, 16
    chan = tmp_path / "data" / "Ex_488_Em_525"
    tiles = {"431040_368180": 0, "431040_394100": 1}
    stacks = {}
    for t, name in enumerate(tiles):
        stack = synth.synthetic_stack(Z, H, W, n_unique=4) + np.uint16(t)
        a = MiniZarrArray.create(str(chan / (name + ".zarr") / "0"), (1, 1, Z, H, W), (1, 1, 4, 32, 32), np.uint16,
                                 compressor="zlib")  # fmt: skip
        a[0, 0] = stack
        stacks[name] = stack
    d = tmp_path / "derivatives"
    d.mkdir()
    dark = np.full((H + 8, W + 8), 90, np.uint16)
    mini_tiff.imwrite(str(d / "DarkMaster_cropped.tif"), dark)
    yy, xx = np.mgrid[0:H, 0:W]
    flats = []
    for side in (0, 1):
        f = (1.0 + 0.2 * side - 0.3 * ((yy - H / 2) / H) ** 2 - 0.2 * ((xx - W / 2) / W) ** 2).astype(np.float32)
        mini_tiff.imwrite(str(d / "flat_{}.tif".format(side)), f)
        flats.append(f)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "channel_rank_worker.py")
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSX_RDZV_DIR", "DSX_FORCE_COMM")}

    def launch(world, results, **extra):
        procs = []
        for r in range(world):
            env = dict(base, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", DSX_RDZV_DIR=str(tmp_path / ("rdzv%d" % world)), **extra)  # fmt: skip
            procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), str(results)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))  # fmt: skip
        outs = []
        for p in procs:
            so, se = p.communicate(timeout=600)
            assert p.returncode == 0, se[-3000:]
            outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
        return sorted(outs, key=lambda o: o["rank"])

    one = launch(1, tmp_path / "r1")
    two = launch(2, tmp_path / "r2")
    forced = launch(1, tmp_path / "r1c", DSX_FORCE_COMM="1")
    names = [n + ".zarr" for n in tiles]
    assert one[0]["done"] == {n: Z for n in names} and one[0]["transport"] == "none"
    assert [o["done"] for o in two] == [{n: Z // 2 for n in names}] * 2  # four output z-chunks: two each
    assert two[0]["transport"] == two[1]["transport"] and two[0]["transport"] in ("host", "rccl")
    assert two[0]["reads"] == ["DarkMaster_cropped.tif", "flat_0.tif", "flat_1.tif"] and two[1]["reads"] == []
    cores = len(os.sched_getaffinity(0))
    assert all(o["io_threads"] == max(2, min(cores // 2, 64)) for o in two)
    # one rank, real communicator: the flat and the dark plane of both tiles went through ncclBroadcast
    assert forced[0]["transport"] == "rccl" and forced[0]["bytes_broadcast"] == 2 * (H * W * 4 + dark.nbytes)
    for name, side in tiles.items():
        got = {}
        for tag in ("r1", "r2", "r1c"):
            out_dir = tmp_path / tag / "destriped_data" / "Ex_488_Em_525" / (name + ".zarr")
            got[tag] = MiniZarrArray.open(str(out_dir / "0"))[0, 0]
            pyr = fo.pyramid(got[tag], 3)
            for lvl in (1, 2):
                np.testing.assert_array_equal(MiniZarrArray.open(str(out_dir / str(lvl)))[0, 0], pyr[lvl])
        np.testing.assert_array_equal(got["r2"], got["r1"])
        np.testing.assert_array_equal(got["r1c"], got["r1"])
        sc = {"retrospective": True, "flatfield": flats[side], "darkfield": dark, "tile_config": None}
        for z in range(Z):
            ref = orc.filter_stripes(stacks[name][z], name, synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, sc, 2500)
            dd = np.abs(got["r2"][z].astype(np.int64) - ref.astype(np.int64))
            assert dd.max() <= 1 and (dd > 0).mean() < 2e-3, (name, z, int(dd.max()))
