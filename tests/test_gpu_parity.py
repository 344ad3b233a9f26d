"""GPU parity tests: HIP engine (through the C ABI) vs the CPU oracle and the golden vectors.

Tolerance: BASELINE.json's north star asks for <= 1e-4 relative (float32) against the reference
NumPy/SciPy path.  Intermediate stages are held to float32 round-off bounds stated per test.
uint16 results go through a truncation, so they may differ by one count where the float value is
within 1e-4 relative of an integer.
"""

import warnings

import numpy as np
import pytest

from aind_smartspim_destripe_amd import engine as eng_mod
from aind_smartspim_destripe_amd import filtering, synth
from oracle import destripe_oracle as orc

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # north star tolerance
# The path has hard decisions (|cH| > threshold, filtering.py:195): a coefficient whose magnitude is
# within float32 round-off (~1e-6 relative) of the threshold can land on the other side than in the
# float64 reference and changes the few pixels under that wavelet's footprint by up to a few percent.
# (The reference's own float32 Zarr path differs from its float64 TIFF path in the same way.)  Such
# flips are rare -- about 0.1 per plane and level -- so parity on large planes is stated as:
# at most FLIP_FRACTION of the pixels may exceed REL_TOL; thresholds and mask counts are checked too.
FLIP_FRACTION = 2e-5
CFGS = {"cells": synth.CELLS_CONFIG, "nocells": synth.NO_CELLS_CONFIG}


def _rel(a, b):
    return np.abs(a.astype(np.float64) - b) / np.abs(b)


def _assert_close(out, ref, what, frac=FLIP_FRACTION):
    """<= frac of the pixels beyond REL_TOL (threshold flips), everything else within REL_TOL."""
    rel = _rel(out, ref)
    n_bad = int((rel > REL_TOL).sum())
    assert n_bad <= max(1, int(frac * rel.size)) if frac > 0 else n_bad == 0, (
        what, n_bad, rel.size, float(rel.max()))
    assert float(np.median(rel)) < 1e-5, (what, float(np.median(rel)))


@pytest.fixture(scope="module")
def engine():
    e = eng_mod.DestripeEngine(0)
    yield e
    e.close()


def _oracle_plane(img, high_int=synth.ZARR_PATH_HIGH_INT):
    which, fore, back = orc.select_config(img, synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, high_int)
    cfg = synth.CELLS_CONFIG if which else synth.NO_CELLS_CONFIG
    out, stages = orc.log_space_fft_filtering(img, return_stages=True, **cfg)
    return which, fore, back, out, stages[::-1]  # stages fine -> coarse, like the engine's level index


@pytest.mark.parametrize("shape", [(64, 64), (128, 96), (256, 256), (101, 103), (512, 512)])
def test_stage_forward_and_thresholds(engine, shape):
    """cH per level (float32 round-off), fg/bg statistic + config choice, Otsu value, threshold."""
    h, w = shape
    planes = synth.synthetic_bank(2, h, w)
    engine.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=2)
    engine.set_stop_after(1)
    try:
        engine.run(planes, out_dtype=np.float32)
        for k in range(2):
            which, fore, back, _, stages = _oracle_plane(planes[k])
            f, b, c = engine.stats(k)
            assert c == which
            assert abs(f - fore) <= 1e-9 * max(1.0, abs(fore)) and abs(b - back) <= 1e-9 * max(1.0, abs(back))
            assert engine.levels == len(stages)
            for lv, st in enumerate(stages):
                ch = engine.level_array(k, lv, eng_mod.STAGE_DETAIL)
                assert ch.shape == st["ch"].shape
                scale = max(1.0, np.abs(st["ch"]).max())
                # float32 analysis of values ~ 2^level * log(pixel): 2e-5 absolute covers 8 levels
                assert np.abs(ch - st["ch"]).max() <= 2e-5 * scale * (2**lv), (shape, k, lv)
                otsu, thr = engine.thresholds(k, lv)
                assert abs(thr - st["threshold"]) <= 2e-5 * max(st["threshold"], 1e-3), (shape, k, lv, thr, st["threshold"])
                assert abs(otsu - st["otsu"]) <= 1e-4 * max(st["otsu"], 1e-6), (shape, k, lv, otsu, st["otsu"])
    finally:
        engine.set_stop_after(0)


@pytest.mark.parametrize("shape", [(64, 64), (128, 96), (256, 256), (512, 512)])
def test_stage_row_filter(engine, shape):
    """Delta_l = ch_filtered - ch per level (mask, exact row median, FFT low-pass with the gain quirk)."""
    h, w = shape
    planes = synth.synthetic_bank(2, h, w)
    engine.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=2)
    engine.set_stop_after(2)
    try:
        engine.run(planes, out_dtype=np.float32)
        for k in range(2):
            _, _, _, _, stages = _oracle_plane(planes[k])
            for lv, st in enumerate(stages):
                delta = engine.level_array(k, lv, eng_mod.STAGE_DETAIL)
                ref = st["ch_filtered"] - st["ch"]
                scale = max(np.abs(st["ch"]).max(), 1e-3)
                bad = np.abs(delta - ref) > 3e-5 * scale * (2**lv)
                # a coefficient within float32 round-off of the threshold may land on the other side
                assert bad.mean() <= 1e-4, (shape, k, lv, int(bad.sum()), float(np.abs(delta - ref).max()))
    finally:
        engine.set_stop_after(0)


def test_golden_small_planes(engine, golden_small):
    """Full outputs vs the vectors the real reference produced (tests/golden/small_full.npz)."""
    g = golden_small
    for case in [str(c) for c in g["cases"]]:
        name, cfg_name, lvl, dt = case.split("__")
        level = None if lvl == "Lmax" else int(lvl[1:])
        img = g[name + "__in"]
        x = img if dt == "u16" else img.astype(np.float32)
        cfg = dict(CFGS[cfg_name])
        cfg["level"] = level
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = filtering.log_space_fft_filtering(x, **cfg)
        ref = g[case + "__out"]
        assert out.shape == ref.shape, case
        rel = _rel(out, ref)
        assert rel.max() < REL_TOL, (case, float(rel.max()))


def test_reference_unit_test_inputs(golden_small):
    """The inputs of the reference's own tests (code/tests/test_filtering.py:151-180)."""
    g = golden_small
    ramp = np.tile(np.linspace(1, 100, 100), (100, 1)).astype(np.float32)
    out = filtering.log_space_fft_filtering(ramp, "db3", 1, 64, 4)
    assert out.shape == ramp.shape and np.all(out > 0)
    assert _rel(out, g["ramp100__L1__out"]).max() < REL_TOL
    out = filtering.log_space_fft_filtering(ramp, "db3", None, 64, 4)
    assert _rel(out, g["ramp100__Lmax__out"]).max() < REL_TOL
    tiny = g["tiny4__in"]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = filtering.log_space_fft_filtering(tiny, wavelet="db3", level=1, sigma=64, max_threshold=4)
        assert any(issubclass(x.category, UserWarning) for x in w)
    assert out.shape == tiny.shape
    assert _rel(out, g["tiny4__L1__out"]).max() < REL_TOL
    # filter_streaks alias
    out2 = filtering.filter_streaks(tiny, wavelet="db3", level=1, sigma=64, max_threshold=4)
    np.testing.assert_array_equal(out, out2)


def test_level0_and_errors():
    img = synth.synthetic_plane(3, 64, 80)
    out = filtering.log_space_fft_filtering(img, level=0)
    np.testing.assert_allclose(out, img.astype(np.float64) + 2.0, rtol=1e-6)
    with pytest.raises(ValueError):
        filtering.log_space_fft_filtering(img, level=1, sigma=0)
    with pytest.raises(ValueError):
        filtering.log_space_fft_filtering(img, level=1, sigma=-3)
    with pytest.raises(ValueError):
        filtering.log_space_fft_filtering(np.zeros((2, 8, 8), np.float32), level=1)


@pytest.mark.parametrize("shape", [(1800, 1800), (1600, 2000), (2048, 2048)])
def test_baseline_shapes_vs_golden(shape, golden_large):
    """BASELINE shapes: filter_stripes semantics with production parameters, both config branches,
    both input dtypes; sampled pixels + plane sum from the real reference."""
    h, w = shape
    name = "s{}".format(h) if h == w else "s{}x{}".format(h, w)
    g = golden_large
    rs = np.random.RandomState(7)
    sy, sx = rs.randint(0, h, 4096), rs.randint(0, w, 4096)
    for k in (0, 1):
        img = synth.synthetic_plane(k, h, w)
        assert int(img.astype(np.uint64).sum()) == int(g["{}__k{}__insum".format(name, k)][0])
        for dt in ("u16", "f32"):
            x = img if dt == "u16" else img.astype(np.float32)
            out, cfg = filtering.destripe_planes(
                x[None], "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, return_config=True, max_batch=1,
            )  # fmt: skip
            key = "{}__k{}__{}".format(name, k, dt)
            assert int(cfg[0]) == int(g[key + "__cfg"][0])
            ref = g[key + "__sample"]
            _assert_close(out[0][sy, sx], ref, key, frac=5e-4)  # 4096 samples: <= 2 under a flipped footprint
            assert abs(out[0].astype(np.float64).sum() - g[key + "__sum"][0]) / g[key + "__sum"][0] < 1e-5


def test_full_plane_2048_vs_oracle():
    """Every pixel of two 2048 x 2048 planes against the CPU oracle."""
    planes = synth.synthetic_bank(2, 2048, 2048)
    out, cfg = filtering.destripe_planes(
        planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
        out_dtype=np.float32, return_config=True, max_batch=2,
    )  # fmt: skip
    for k in range(2):
        which, _, _, ref, _ = _oracle_plane(planes[k])
        assert int(cfg[k]) == which
        _assert_close(out[k], ref, ("2048", k))


@pytest.mark.parametrize("shape", [(300, 2048), (258, 2047), (2048, 1024), (640, 2046)])
def test_wide_planes_vs_oracle(shape):
    """2048-wide (and nearly so) planes of other heights: the row filter's compile-time instantiations
    (1026 = 19*9*6 direct, 515 embedded in 1071) serve any plane whose level-1 / level-2 widths match, and
    neighbouring widths fall back to the generic kernels."""
    planes = np.stack([synth.synthetic_plane(k, *shape) for k in (0, 1)])
    out, cfg = filtering.destripe_planes(
        planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
        out_dtype=np.float32, return_config=True, max_batch=2,
    )  # fmt: skip
    for k in range(2):
        which, _, _, ref, _ = _oracle_plane(planes[k])
        assert int(cfg[k]) == which
        assert out[k].shape == ref.shape
        _assert_close(out[k], ref, (shape, k), frac=1e-4)


def test_uint16_output_and_cohorts():
    """uint16 result (clip + truncate) and n > max_batch (several cohorts, ragged last one)."""
    planes = synth.synthetic_bank(7, 128, 160)
    out_f = filtering.destripe_planes(planes, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                      synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, max_batch=8)  # fmt: skip
    out_u = filtering.destripe_planes(planes, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                      synth.ZARR_PATH_HIGH_INT, out_dtype=np.uint16, max_batch=3)  # fmt: skip
    assert out_u.dtype == np.uint16 and out_u.shape == out_f.shape
    np.testing.assert_array_equal(out_u, np.clip(out_f, 0, 65535).astype(np.uint16))
    for k in range(7):
        ref = orc.filter_stripes(planes[k], "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                 synth.ZARR_PATH_HIGH_INT)  # fmt: skip
        assert _rel(out_f[k], ref).max() < REL_TOL
        d = np.abs(out_u[k].astype(np.int64) - np.clip(ref, 0, 65535).astype(np.uint16).astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3
    # empty batch
    empty = filtering.destripe_planes(planes[:0], "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                      synth.ZARR_PATH_HIGH_INT)  # fmt: skip
    assert empty.shape == (0, 128, 160)


def test_filter_stripes_api_and_shading(golden_misc):
    """filter_stripes: float64 without shading, uint16 with retrospective shading (golden vectors),
    per-hemisphere flat lookup and its KeyError."""
    g = golden_misc
    flat, dark = g["shade__flat"], g["shade__dark"]
    sc = {"retrospective": True, "flatfield": flat, "darkfield": dark, "tile_config": {}}
    for k in (0, 1):
        img = g["shade__k{}__in".format(k)]
        out = filtering.filter_stripes(img, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, sc,
                                       synth.ZARR_PATH_HIGH_INT)  # fmt: skip
        ref = g["shade__k{}__out".format(k)]
        assert out.dtype == np.uint16 and out.shape == ref.shape
        d = np.abs(out.astype(np.int64) - ref.astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (k, int(d.max()), float((d > 0).mean()))
        plain = filtering.filter_stripes(img, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                         synth.ZARR_PATH_HIGH_INT)  # fmt: skip
        assert plain.dtype == np.float64
        ref_plain = orc.filter_stripes(img, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                       synth.ZARR_PATH_HIGH_INT)  # fmt: skip
        assert _rel(plain, ref_plain).max() < REL_TOL
    # prospective flats: list indexed by brain side through the tile config
    img = g["shade__k0__in"]
    sc2 = {"retrospective": False, "flatfield": [flat * 2.0, flat], "darkfield": dark,
           "tile_config": {"X1": {"Y1": 1}}}  # fmt: skip
    out = filtering.filter_stripes(img, "X1_Y1", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, sc2,
                                   synth.ZARR_PATH_HIGH_INT)  # fmt: skip
    d = np.abs(out.astype(np.int64) - g["shade__k0__out"].astype(np.int64))
    assert d.max() <= 1
    with pytest.raises(KeyError):
        filtering.filter_stripes(img, "X3_Y1", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, sc2)
    with pytest.raises(ValueError):
        bad = {"retrospective": True, "flatfield": flat[:-1], "darkfield": dark, "tile_config": {}}
        filtering.filter_stripes(img, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, bad)
    with pytest.raises(ValueError):
        bad = {"retrospective": True, "flatfield": flat, "darkfield": dark[:50], "tile_config": {}}
        filtering.filter_stripes(img, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, bad)


def test_config_decision_edges(engine):
    """Decision rule of filtering.py:462 incl. empty classes and the 2700 / 2500 thresholds."""
    h, w = 64, 64
    base = np.full((h, w), 100, np.uint16)
    cases = []
    a = base.copy(); cases.append(a)                       # no foreground at all -> no cells
    a = base.copy(); a[:8, :8] = 2600; cases.append(a)     # fg mean 2600
    a = np.full((h, w), 3000, np.uint16); cases.append(a)  # all foreground: back mean 0.0
    a = base.copy(); a[0, 0] = 384; cases.append(a)        # one pixel exactly at the cut-off
    a = base.copy(); a[0, 0] = 383; cases.append(a)
    planes = np.stack(cases)
    for high in (2500, 2700):
        engine.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, high, max_batch=8)
        _, cfg = engine.run(planes, out_dtype=np.float32, return_cfg=True)
        for k, p in enumerate(planes):
            which, fore, back = orc.select_config(p, synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, high)
            f, b, c = engine.stats(k)
            assert (c, f, b) == (which, fore, back), (high, k)
            assert int(cfg[k]) == which
    # fractional float32 pixels around the float16 cut-off
    fr = np.linspace(380.0, 388.0, h * w).astype(np.float32).reshape(h, w)
    engine.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, 100.0, max_batch=1)
    engine.run(fr[None], out_dtype=np.float32)
    fore, back, _ = orc.get_foreground_background_mean(fr)
    f, b, c = engine.stats(0)
    assert abs(f - float(fore)) < 1e-3 and abs(b - float(back)) < 1e-3


def test_idempotent_and_batch_invariant():
    """Same plane in different batch positions / batch sizes gives bit-identical results."""
    planes = synth.synthetic_bank(4, 256, 192)
    a = filtering.destripe_planes(planes, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, 2500,
                                  out_dtype=np.float32, max_batch=4)  # fmt: skip
    b = filtering.destripe_planes(planes[::-1].copy(), "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                  2500, out_dtype=np.float32, max_batch=2)  # fmt: skip
    np.testing.assert_array_equal(a, b[::-1])


def test_mixed_levels_between_configs(engine):
    """cells_config and no_cells_config with different depths: levels above a config's depth get Delta = 0."""
    h, w = 128, 128
    cells = dict(synth.CELLS_CONFIG, level=2)
    nocells = dict(synth.NO_CELLS_CONFIG, level=4)
    planes = synth.synthetic_bank(2, h, w)
    engine.plan(h, w, cells, nocells, synth.ZARR_PATH_HIGH_INT, max_batch=2)
    out, cfg = engine.run(planes, out_dtype=np.float32, return_cfg=True)
    for k in range(2):
        ref = orc.filter_stripes(planes[k], "t", nocells, cells, None, synth.ZARR_PATH_HIGH_INT)
        assert _rel(out[k], ref).max() < REL_TOL, (k, int(cfg[k]))


def test_mask_flip_accounting_2048(engine):
    """Hard decisions on full-size planes: per level, the number of masked coefficients (Delta == 0
    marks a masked one) may differ from the float64 oracle only by the handful of coefficients that
    sit within float32 round-off of the threshold."""
    planes = synth.synthetic_bank(2, 2048, 2048, first=1)
    engine.plan(2048, 2048, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=2)
    engine.set_stop_after(2)
    try:
        engine.run(planes, out_dtype=np.float32)
        for k in range(2):
            _, _, _, _, stages = _oracle_plane(planes[k])
            for lv, st in enumerate(stages):
                delta = engine.level_array(k, lv, eng_mod.STAGE_DETAIL)
                mask_ref = np.abs(st["ch"]) > st["threshold"]
                mask_gpu = delta == 0.0
                # an unmasked coefficient has Delta == 0 only by coincidence: count disagreements both ways
                flips = int((mask_gpu != mask_ref).sum())
                assert flips <= max(3, int(2e-5 * mask_ref.size)), (k, lv, flips, int(mask_ref.sum()))
                otsu, thr = engine.thresholds(k, lv)
                assert abs(thr - st["threshold"]) <= 2e-5 * max(st["threshold"], 1e-3)
    finally:
        engine.set_stop_after(0)


def test_float32_input_2048_and_properties():
    """Zarr-path dtype (float32 planes holding integers) at full size, plus size-independent properties:
    float32 and uint16 planes with the same pixels give bit-identical results, and a constant plane has
    nothing to filter (result x + 2)."""
    planes = synth.synthetic_bank(2, 2048, 2048).astype(np.float32)
    out, cfg = filtering.destripe_planes(planes, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                         synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, return_config=True,
                                         max_batch=2)  # fmt: skip
    out_u = filtering.destripe_planes(planes.astype(np.uint16), "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                      synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, max_batch=2)  # fmt: skip
    np.testing.assert_array_equal(out, out_u)  # same pixels, same arithmetic on the device
    for k in range(2):
        ref = orc.filter_stripes(planes[k], "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                 synth.ZARR_PATH_HIGH_INT)  # fmt: skip
        _assert_close(out[k], ref, ("f32-2048", k))
    const = np.full((1, 2048, 2048), 777, np.uint16)
    res = filtering.destripe_planes(const, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                    synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, max_batch=1)  # fmt: skip
    assert np.abs(res - 779.0).max() < 779.0 * 2e-5
    res_u = filtering.destripe_planes(const, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                      synth.ZARR_PATH_HIGH_INT, out_dtype=np.uint16, max_batch=1)  # fmt: skip
    assert int(np.abs(res_u.astype(np.int64) - 779).max()) <= 1
