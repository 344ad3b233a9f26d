"""Wavelets other than db3 (reference: ``pywt.wavedec2(x, wavelet=...)``, filtering.py:176, 221).

``tests/golden/wavelets.npz`` holds results of the REAL reference (``oracle/make_golden_wavelets.py``) for
20 wavelet / shape / dtype / level combinations; ``aind_smartspim_destripe_amd/wavelet_table.npz`` holds the filter
banks printed by PyWavelets 1.1.1 (``oracle/make_wavelet_table.py``).  CPU: the table and the oracle against those
vectors; GPU (``-m gpu``): the tap-count-generic kernels (``csrc/dsx_wavelet.h``) through the C ABI.
"""

import json
import os

import numpy as np
import pytest

from aind_smartspim_destripe_amd import synth, wavelets
from oracle import destripe_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wavelets.npz")


@pytest.fixture(scope="module")
def golden_wavelets():
    with np.load(GOLDEN) as z:
        return {k: z[k] for k in z.files}


def _cases(g):
    return json.loads(str(g["cases"]))


def _input(c):
    img = synth.synthetic_plane(c["k"], c["H"], c["W"])
    return img if c["dtype"] == "u16" else img.astype(np.float32)


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_table_matches_the_db3_constants_and_is_consistent():
    names = wavelets.wavelist()
    assert len(names) == 106 and {"haar", "db3", "sym4", "coif17", "bior6.8", "rbio3.9", "dmey"} <= set(names)
    dec_lo, dec_hi, rec_lo, rec_hi = wavelets.filter_bank("db3")
    np.testing.assert_array_equal(dec_lo, orc.DB3_DEC_LO)
    np.testing.assert_array_equal(dec_hi, orc.DB3_DEC_HI)
    np.testing.assert_array_equal(rec_lo, orc.DB3_REC_LO)
    np.testing.assert_array_equal(rec_hi, orc.DB3_REC_HI)
    for n in names:
        if n == "dmey":
            continue
        bank = wavelets.filter_bank(n)
        assert len(bank[0]) % 2 == 0 and 2 <= len(bank[0]) <= wavelets.MAX_TAPS, n
        # perfect reconstruction: sum_k rec_lo[k] dec_lo[2 m + (F - 1) - k] + rec_hi[..] dec_hi[..] = delta(m)
        full = np.convolve(bank[2], bank[0]) + np.convolve(bank[3], bank[1])
        mid = len(full) // 2
        np.testing.assert_allclose(full[mid], 2.0, atol=1e-7, err_msg=n)
        np.testing.assert_allclose(np.delete(full, mid)[(mid + 1) % 2 :: 2], 0.0, atol=1e-7, err_msg=n)
    assert wavelets.filter_length("coif17") == 102 and wavelets.filter_length("haar") == 2


def test_unknown_and_unsupported_names_raise_value_error():
    with pytest.raises(ValueError, match="Unknown wavelet name 'db99'"):
        wavelets.filter_bank("db99")
    with pytest.raises(ValueError, match="dmey"):
        wavelets.filter_bank("dmey")
    with pytest.raises(ValueError):
        wavelets.filter_bank((np.ones(3), np.ones(3), np.ones(3), np.ones(3)))  # odd length


def test_oracle_against_the_reference_for_every_wavelet_case(golden_wavelets):
    g = golden_wavelets
    cases = _cases(g)
    assert len(cases) == 20
    for c in cases:
        bank = wavelets.filter_bank(c["wavelet"])
        out, stages = orc.log_space_fft_filtering(_input(c), wavelet=bank, level=c["level"], sigma=c["sigma"],
                                                  max_threshold=c["max_threshold"], return_stages=True)
        ref = g[c["key"] + "__out"]
        assert out.shape == ref.shape and out.dtype == ref.dtype, c["key"]
        assert [tuple(s["ch"].shape) for s in stages] == [tuple(x) for x in g[c["key"] + "__chshape"]], c["key"]
        tol = 1e-11 if c["dtype"] == "u16" else 2e-5
        assert _rel(out, ref) < tol, (c["key"], _rel(out, ref))
        if c["dtype"] == "u16":
            np.testing.assert_allclose([s["otsu"] for s in stages], g[c["key"] + "__otsu"], rtol=1e-10)


def test_max_level_follows_the_filter_length():
    assert orc.dwt_max_level(200, 2) == 7 and orc.dwt_max_level(5, 6) == 0 and orc.dwt_max_level(4, 6) == 0
    from aind_smartspim_destripe_amd import filtering

    assert filtering._max_level((64, 80), 2) == 6 and filtering._max_level((96, 120), 40) == 1
    assert filtering._max_level((2048, 2048)) == 8


# ---------------------------------------------------------------------------------------------------------
# GPU: the generic level kernels through the C ABI
# ---------------------------------------------------------------------------------------------------------
MAX_FLIPS = lambda size: max(3, int(2e-5 * size))  # noqa: E731  (as tests/test_gpu_parity.py)


def _gpu_case(engine, x, name, level, sigma, max_thr, what, ref=None):
    """One plane, one config: engine result under the parity statement of tests/parity_util.py."""
    from parity_util import check_plane, gpu_deltas

    cfg = {"wavelet": name, "level": level, "sigma": sigma, "max_threshold": max_thr}
    ocfg = dict(cfg, wavelet=wavelets.filter_bank(name))  # the oracle takes the bank itself
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        deltas = gpu_deltas(engine, x[None], high_int=2500, cells=cfg, nocells=cfg)
        out = engine.run(x[None], out_dtype=np.float32)[0].astype(np.float64)
        check_plane(out, x, deltas[0], what, ocfg, MAX_FLIPS, ref=ref)
    return out


@pytest.mark.gpu
def test_gpu_every_wavelet_case_against_the_reference(golden_wavelets):
    """The 20 reference runs of tests/golden/wavelets.npz: shapes, every pixel within 1e-4 (parity_util)."""
    from aind_smartspim_destripe_amd import engine as eng_mod

    g = golden_wavelets
    e = eng_mod.DestripeEngine(0)
    try:
        for c in _cases(g):
            if c["wavelet"] == "db3":
                continue  # has its own test below (the name alone selects the specialised kernels)
            ref = g[c["key"] + "__out"]
            out = _gpu_case(e, _input(c), c["wavelet"], c["level"], c["sigma"], c["max_threshold"], c["key"], ref=ref)
            assert out.shape == ref.shape, c["key"]
            assert [tuple(e.level_shape(lv)) for lv in range(e.levels)][::-1] == [tuple(s) for s in g[c["key"] + "__chshape"]]
    finally:
        e.close()


@pytest.mark.gpu
def test_gpu_db3_through_the_generic_kernels_matches_the_specialised_ones(golden_wavelets, monkeypatch):
    """DSX_GENERIC_DB3=1 hands db3 over as a filter bank: same reference values, and the two kernel families agree
    to float32 round-off on a plane with both edge parities."""
    from aind_smartspim_destripe_amd import engine as eng_mod

    g = golden_wavelets
    c = [c for c in _cases(g) if c["wavelet"] == "db3"][0]
    x = _input(c)
    e = eng_mod.DestripeEngine(0)
    try:
        fast = _gpu_case(e, x, "db3", c["level"], c["sigma"], c["max_threshold"], "db3 specialised", ref=g[c["key"] + "__out"])
        monkeypatch.setenv("DSX_GENERIC_DB3", "1")
        gen = _gpu_case(e, x, "db3", c["level"], c["sigma"], c["max_threshold"], "db3 generic", ref=g[c["key"] + "__out"])
        monkeypatch.delenv("DSX_GENERIC_DB3")
        assert _rel(gen, fast) < 2e-5
        for shape in ((200, 232), (131, 77)):
            planes = synth.synthetic_bank(3, *shape)
            e.plan(shape[0], shape[1], synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, 2500, max_batch=4)
            a, ca = e.run(planes, out_dtype=np.float32, return_cfg=True)
            monkeypatch.setenv("DSX_GENERIC_DB3", "1")
            e.plan(shape[0], shape[1], synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, 2500, max_batch=4)
            b, cb = e.run(planes, out_dtype=np.float32, return_cfg=True)
            monkeypatch.delenv("DSX_GENERIC_DB3")
            np.testing.assert_array_equal(ca, cb)
            r = np.abs(a.astype(np.float64) - b) / np.abs(a)
            assert np.median(r) < 1e-6 and (r > 1e-4).mean() < 1e-3, (shape, float(r.max()))
    finally:
        e.close()


@pytest.mark.gpu
def test_gpu_wavelet_batch_configs_statistic_shading_and_casts():
    """A batch through the generic kernels with the production config pair renamed to sym4: per-plane config choice
    (fg/bg statistic fused into the level-1 kernel), both configs' results, uint16 cast, shading epilogue, and the
    reference's API (``filter_stripes`` / ``log_space_fft_filtering`` / ``filter_streaks``) with a wavelet name."""
    from aind_smartspim_destripe_amd import engine as eng_mod
    from aind_smartspim_destripe_amd import filtering
    from parity_util import check_plane, gpu_deltas

    name = "sym4"
    bank = wavelets.filter_bank(name)
    cells, nocells = dict(synth.CELLS_CONFIG, wavelet=name), dict(synth.NO_CELLS_CONFIG, wavelet=name)
    ocells, onocells = dict(cells, wavelet=bank), dict(nocells, wavelet=bank)
    h, w = 203, 232
    planes = synth.synthetic_bank(5, h, w)  # plane 0 and 4 carry cells
    e = eng_mod.DestripeEngine(0)
    try:
        deltas = gpu_deltas(e, planes, high_int=2500, cells=cells, nocells=nocells)
        out, used = e.run(planes, out_dtype=np.float32, return_cfg=True)
        out16 = e.run(planes, out_dtype=np.uint16)
        assert out.shape == (5, h + 1, w)
        seen = set()
        for k in range(5):
            which, fore, back = orc.select_config(planes[k], onocells, ocells, 2500)
            f, b, cfg_k = e.stats(k)
            assert (cfg_k, f, b) == (which, fore, back) and int(used[k]) == which
            seen.add(which)
            check_plane(out[k].astype(np.float64), planes[k], deltas[k], "sym4 plane %d" % k, ocells if which else onocells, MAX_FLIPS)
            d = np.abs(out16[k].astype(np.int64) - np.minimum(out[k], 65535.0).astype(np.int64))
            assert d.max() == 0
        assert seen == {0, 1}
        with pytest.raises(ValueError, match="same wavelet"):
            e.plan(h, w, cells, synth.NO_CELLS_CONFIG, 2500, max_batch=1)
        with pytest.raises(ValueError, match="Unknown wavelet name"):
            e.plan(h, w, dict(cells, wavelet="nope"), dict(nocells, wavelet="nope"), 2500, max_batch=1)
    finally:
        e.close()
    # the two configs name DIFFERENT wavelets: the host layer decides per plane first (dsx_foreground_background), then
    # runs every plane through an engine of the bank its config names -- the reference's order (filtering.py:459-467)
    mixed_cells, mixed_nocells = dict(synth.CELLS_CONFIG, wavelet="sym4"), dict(synth.NO_CELLS_CONFIG, wavelet="db3")
    got, used = filtering.destripe_planes(planes, "t", mixed_nocells, mixed_cells, None, 2500, out_dtype=np.float32,
                                          return_config=True)
    assert set(int(u) for u in used) == {0, 1}
    for k in range(5):
        which, _, _ = orc.select_config(planes[k], mixed_nocells, dict(mixed_cells, wavelet=bank), 2500)
        assert int(used[k]) == which
        ref = orc.filter_stripes(planes[k], "t", mixed_nocells, dict(mixed_cells, wavelet=bank), None, 2500)
        rel = np.abs(got[k] - ref) / np.abs(ref)
        assert (rel > 1e-4).mean() < 2e-3 and rel.max() < 0.5, (k, float(rel.max()))
    one = filtering.filter_stripes(planes[0], "t", mixed_nocells, mixed_cells, None, 2500)
    np.testing.assert_array_equal(one.astype(np.float32), got[0])
    # reference API with a wavelet name, float32 plane, even shape; shading epilogue vs the oracle's
    x = synth.synthetic_plane(3, 96, 128)
    rng = np.random.RandomState(5)
    flat = (0.8 + 0.4 * rng.rand(96, 128)).astype(np.float32)
    dark = (90 + 20 * rng.rand(100, 130)).astype(np.float32)
    sc = {"retrospective": True, "flatfield": flat, "darkfield": dark, "tile_config": {}}
    got = filtering.filter_stripes(x, "X_0_Y_0", nocells, cells, sc, 2500)
    ref = orc.filter_stripes(x, "X_0_Y_0", onocells, ocells, sc, 2500)
    assert got.dtype == np.uint16 and got.shape == ref.shape
    d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3, (int(d.max()), float((d > 0).mean()))
    a = filtering.log_space_fft_filtering(x.astype(np.float32), wavelet="bior2.2", level=2, sigma=64, max_threshold=4)
    b = filtering.filter_streaks(x.astype(np.float32), wavelet="bior2.2", level=2, sigma=64, max_threshold=4)
    r = orc.log_space_fft_filtering(x.astype(np.float32), wavelet=wavelets.filter_bank("bior2.2"), level=2, sigma=64, max_threshold=4)
    np.testing.assert_array_equal(a, b)
    assert a.dtype == np.float64 and _rel(a, r) < 1e-4
    # level 0 is wavelet-independent: image + 2
    z = filtering.log_space_fft_filtering(x, wavelet="coif2", level=0, sigma=64, max_threshold=4)
    np.testing.assert_allclose(z, x.astype(np.float64) + 2.0, rtol=1e-6)


def test_rbio31_float32_spread_of_the_reference_algorithm_itself():
    """rbio3.1 (4 taps, dec_lo = [-0.35, 1.06, 1.06, -0.35]) is the one PyWavelets bank whose synthesis amplifies float32
    round-off beyond the 1e-4 of the parity statement at full depth: the REFERENCE ALGORITHM's own float32 regime (float32
    planes, the Zarr path) and float64 regime (uint16 planes) differ by 1e-3 on a 260 x 331 plane, every other bank by
    < 3e-5 (the few with a larger MAXIMUM -- coif1, rbio3.9 here -- differ by a threshold decision, with the usual median).
    CPU: the oracle in both regimes."""
    import warnings

    img = synth.synthetic_plane(3, 196, 251)
    spread = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name in ("rbio3.1", "bior3.1", "db3"):
            bank = wavelets.filter_bank(name)
            a = orc.log_space_fft_filtering(img, wavelet=bank, level=None, sigma=128, max_threshold=12)
            b = orc.log_space_fft_filtering(img.astype(np.float32), wavelet=bank, level=None, sigma=128, max_threshold=12)
            r = np.abs(a - b) / np.abs(a)
            spread[name] = (float(r.max()), float(np.median(r)))
    assert spread["rbio3.1"][0] > 3e-4 and spread["rbio3.1"][1] > 4e-6, spread
    for name in ("bior3.1", "db3"):
        assert spread[name][0] < 5e-5 and spread[name][1] < 4e-6, (name, spread)


@pytest.mark.gpu
def test_gpu_rbio31_within_the_reference_s_own_regime_spread():
    """The engine on rbio3.1 at full depth: median as for every other bank, maximum within 5e-3 (the reference's own two
    regimes are 1e-3 apart, see the CPU test above); at depth 2 the 1e-4 statement holds as usual."""
    from aind_smartspim_destripe_amd import filtering

    img = synth.synthetic_plane(3, 260, 331)
    bank = wavelets.filter_bank("rbio3.1")
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = filtering.log_space_fft_filtering(img.astype(np.float32), wavelet="rbio3.1", level=None, sigma=128, max_threshold=12)
        ref = orc.log_space_fft_filtering(img.astype(np.float32), wavelet=bank, level=None, sigma=128, max_threshold=12)
        r = np.abs(got - ref) / np.abs(ref)
        assert float(np.median(r)) < 2e-5 and float(r.max()) < 5e-3, (float(np.median(r)), float(r.max()))
        got2 = filtering.log_space_fft_filtering(img.astype(np.float32), wavelet="rbio3.1", level=2, sigma=128, max_threshold=12)
        ref2 = orc.log_space_fft_filtering(img.astype(np.float32), wavelet=bank, level=2, sigma=128, max_threshold=12)
        assert _rel(got2, ref2) < 1e-4
