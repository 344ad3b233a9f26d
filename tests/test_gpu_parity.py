"""GPU parity tests: HIP engine (through the C ABI) vs the CPU oracle and the golden vectors.

Tolerance: BASELINE.json's north star asks for <= 1e-4 relative (float32) against the reference
NumPy/SciPy path.  The statement the tests prove is in ``tests/parity_util.py``: every pixel within
1e-4, except pixels in the footprint of a coefficient whose hard mask decision (|cH| > threshold)
fell the other way than in the oracle -- each such pixel is shown to lie under a flipped coefficient,
is capped at 5e-2, and the flips per level are counted and bounded.  Intermediate stages are held to
float32 round-off bounds stated per test.  uint16 results go through a truncation, so they may
differ by one count where the float value is within 1e-4 relative of an integer.
"""

import os
import warnings

import numpy as np
import pytest

from aind_smartspim_destripe_amd import engine as eng_mod
from aind_smartspim_destripe_amd import filtering, synth
from oracle import destripe_oracle as orc
from parity_util import REL_TOL, check_plane, gpu_deltas, oracle_plane, rel_err

pytestmark = pytest.mark.gpu

# flips per plane and level the tests accept: coefficients within float32 round-off (~1e-6 relative) of
# the threshold; observed 0-2 per level at 2048 x 2048 (printed by the tests)
MAX_FLIPS = lambda size: max(3, int(2e-5 * size))  # noqa: E731
CFGS = {"cells": synth.CELLS_CONFIG, "nocells": synth.NO_CELLS_CONFIG}

_rel = rel_err
_oracle_plane = oracle_plane


def _check_plane(out, img, deltas, what, pos=None, ref=None, stages=None, which=None):
    """One plane against the oracle / golden values, every deviation accounted for (tests/parity_util.py)."""
    if which is None:
        which, _, _ = orc.select_config(img, synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT)
    cfg = synth.CELLS_CONFIG if which else synth.NO_CELLS_CONFIG
    return check_plane(out, img, deltas, what, cfg, MAX_FLIPS, ref=ref, stages=stages, pos=pos)


@pytest.fixture(scope="module")
def engine():
    e = eng_mod.DestripeEngine(0)
    yield e
    e.close()


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


def test_stack_mode_3d_input_vs_golden():
    """The reference's 3-D input mode (filtering.py:182-183, 188, 210-211): ONE Otsu threshold per level for the whole
    stack.  16 runs of the real reference (tests/golden/stack3d.npz, oracle/make_golden_stack3d.py): two stacks (one
    with odd planes), both configs, full depth and level 2, uint16 and float32 input.  A plane of the stack carries a
    bright block, so its coefficients set the histogram range of every plane: filtering the planes independently gives
    another result (checked), the stack mode must give the reference's."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stack3d.npz"), allow_pickle=False)
    differs = 0
    for case in [str(c) for c in g["cases"]]:
        name, cfg_name, lvl, dt = case.split("__")
        x = g[name + "__in"] if dt == "u16" else g[name + "__in"].astype(np.float32)
        cfg = dict(CFGS[cfg_name])
        cfg["level"] = None if lvl == "Lmax" else int(lvl[1:])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = filtering.log_space_fft_filtering(x, **cfg)
            alone = np.stack([filtering.log_space_fft_filtering(p, **cfg) for p in x])
        ref = g[case + "__out"]
        assert out.shape == ref.shape and out.dtype == np.float64, case
        rel = _rel(out, ref)
        assert rel.max() < REL_TOL, (case, float(rel.max()))
        differs += int(_rel(alone, ref).max() > 10 * REL_TOL)
    assert differs >= 8  # the shared threshold matters on these stacks
    # the stack must fit one cohort, and the mode is off again afterwards
    e = eng_mod.DestripeEngine(0)
    try:
        e.plan(64, 96, synth.CELLS_CONFIG, synth.CELLS_CONFIG, 2700, max_batch=2)
        e.set_stack_mode(True)
        with pytest.raises(eng_mod.DsxError, match="stack mode"):
            e.run(g["a__in"], out_dtype=np.float32)
    finally:
        e.close()


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
        filtering.log_space_fft_filtering(np.zeros((2, 2, 8, 8), np.float32), level=1)


@pytest.mark.parametrize("shape", [(1800, 1800), (1600, 2000), (2048, 2048)])
def test_baseline_shapes_vs_golden(engine, shape, golden_large):
    """BASELINE shapes: filter_stripes semantics with production parameters, both config branches,
    both input dtypes; sampled pixels + plane sum + per-level thresholds from the real reference.
    Samples beyond 1e-4 must lie under a coefficient whose mask bit differs from the oracle's (run in
    the same dtype regime as the golden vector)."""
    h, w = shape
    name = "s{}".format(h) if h == w else "s{}x{}".format(h, w)
    g = golden_large
    rs = np.random.RandomState(7)
    sy, sx = rs.randint(0, h, 4096), rs.randint(0, w, 4096)
    planes = synth.synthetic_bank(2, h, w)
    deltas = gpu_deltas(engine, planes)
    thr_gpu = [[engine.thresholds(k, lv)[1] for lv in range(engine.levels)] for k in (0, 1)]
    for k in (0, 1):
        img = planes[k]
        assert int(img.astype(np.uint64).sum()) == int(g["{}__k{}__insum".format(name, k)][0])
        for dt in ("u16", "f32"):
            x = img if dt == "u16" else img.astype(np.float32)
            out, cfg = filtering.destripe_planes(
                x[None], "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, return_config=True, max_batch=1,
            )  # fmt: skip
            key = "{}__k{}__{}".format(name, k, dt)
            assert int(cfg[0]) == int(g[key + "__cfg"][0])
            # thresholds of the real reference, coarse -> fine in the fixture
            np.testing.assert_allclose(thr_gpu[k][::-1], g[key + "__thr"], rtol=3e-5)
            _, _, _, _, stages = oracle_plane(x)
            _check_plane(out[0], x, deltas[k], key, pos=(sy, sx), ref=g[key + "__sample"], stages=stages)
            assert abs(out[0].astype(np.float64).sum() - g[key + "__sum"][0]) / g[key + "__sum"][0] < 1e-5


def test_full_plane_2048_vs_oracle(engine):
    """Every pixel of two 2048 x 2048 planes against the CPU oracle."""
    planes = synth.synthetic_bank(2, 2048, 2048)
    deltas = gpu_deltas(engine, planes)
    out, cfg = filtering.destripe_planes(
        planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
        out_dtype=np.float32, return_config=True, max_batch=2,
    )  # fmt: skip
    for k in range(2):
        which, _, _, ref, stages = oracle_plane(planes[k])
        assert int(cfg[k]) == which
        _check_plane(out[k], planes[k], deltas[k], ("2048", k), ref=ref, stages=stages)


@pytest.mark.parametrize("shape", [(300, 2048), (258, 2047), (2048, 1024), (640, 2046)])
def test_wide_planes_vs_oracle(engine, shape):
    """2048-wide (and nearly so) planes of other heights: the row filter's compile-time instantiations
    (1026 = 19*9*6 direct, 515 embedded in 1071) serve any plane whose level-1 / level-2 widths match, and
    neighbouring widths fall back to the generic kernels."""
    planes = np.stack([synth.synthetic_plane(k, *shape) for k in (0, 1)])
    deltas = gpu_deltas(engine, planes)
    out, cfg = filtering.destripe_planes(
        planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
        out_dtype=np.float32, return_config=True, max_batch=2,
    )  # fmt: skip
    for k in range(2):
        which, _, _, ref, stages = oracle_plane(planes[k])
        assert int(cfg[k]) == which
        assert out[k].shape == ref.shape
        _check_plane(out[k], planes[k], deltas[k], (shape, k), ref=ref, stages=stages)


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


def test_float32_input_2048_and_properties(engine):
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
    deltas = gpu_deltas(engine, planes)
    for k in range(2):
        _, _, _, ref, stages = oracle_plane(planes[k])  # float32 regime of the reference (Zarr path)
        _check_plane(out[k], planes[k], deltas[k], ("f32-2048", k), ref=ref, stages=stages)
    const = np.full((1, 2048, 2048), 777, np.uint16)
    res = filtering.destripe_planes(const, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                    synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, max_batch=1)  # fmt: skip
    assert np.abs(res - 779.0).max() < 779.0 * 2e-5
    res_u = filtering.destripe_planes(const, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                      synth.ZARR_PATH_HIGH_INT, out_dtype=np.uint16, max_batch=1)  # fmt: skip
    assert int(np.abs(res_u.astype(np.int64) - 779).max()) <= 1


# ---- the configuration bench.py times: one cohort split over the engine's sub-cohort streams ---------------
def _engine_with_streams(n):
    old = os.environ.get("DSX_STREAMS")
    os.environ["DSX_STREAMS"] = str(n)
    try:
        return eng_mod.DestripeEngine(0)  # dsx_init reads DSX_STREAMS
    finally:
        if old is None:
            del os.environ["DSX_STREAMS"]
        else:
            os.environ["DSX_STREAMS"] = old


@pytest.mark.parametrize(
    "n,shape,n_unique",
    [
        (64, (2048, 2048), 8),   # 4 parts of 16 planes: the split bench.py runs (256 planes -> 4 x 64)
        (128, (512, 512), 16),   # 4 parts of 32 planes, 6 levels
        (68, (640, 640), 17),    # 4 parts of 17 planes, 7 levels: control-block slices that are not
                                 # 16-byte aligned (k_zero3's fallback branch), ragged last part
        (64, (1800, 1800), 6),   # BASELINE configs[4]: non-power-of-two tile, embedded mixed-radix plans (1815, 960)
        (64, (1600, 2000), 6),   # the production tile (zarr_destriper.py:1256 of the reference), plans 2048 / 1024
    ],
)
def test_multistream_cohort(n, shape, n_unique, golden_large):
    """A cohort of >= 32 planes is split into parts that run their launch chains on separate HIP streams
    (dsx.hip run_cohort_split: fork / join events, per-part slices of the workspace and control block).
    Planes from EVERY part are compared with the oracle, the first two planes at 2048 x 2048 with the
    reference's golden samples, and the whole result must be bit-identical to a single-stream engine."""
    h, w = shape
    stack = synth.synthetic_stack(n, h, w, n_unique=n_unique)
    e4 = _engine_with_streams(4)
    e1 = _engine_with_streams(1)
    try:
        e4.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=n)
        e1.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=n)
        out4, cfg4 = e4.run(stack, out_dtype=np.float32, return_cfg=True)
        out1, cfg1 = e1.run(stack, out_dtype=np.float32, return_cfg=True)
        np.testing.assert_array_equal(cfg4, cfg1)
        assert np.array_equal(out4, out1), "sub-cohort streams change the result"
        u4 = e4.run(stack, out_dtype=np.uint16)
        np.testing.assert_array_equal(u4, np.clip(out4, 0, 65535).astype(np.uint16))
        # one plane from every part (parts are contiguous runs of ceil(n / 4) planes) + the last plane
        per = (n + 3) // 4
        picks = sorted({0, per + 1, 2 * per + 3, 3 * per + per // 2, n - 1})
        sub = np.ascontiguousarray(stack[picks])
        deltas = gpu_deltas(e1, sub)
        for i, z in enumerate(picks):
            which, _, _, ref, stages = oracle_plane(stack[z])
            assert int(cfg4[z]) == which, (z, int(cfg4[z]), which)
            _check_plane(out4[z], stack[z], deltas[i], ("multistream", shape, z), ref=ref, stages=stages)
        if shape == (2048, 2048):
            g = golden_large
            rs = np.random.RandomState(7)
            sy, sx = rs.randint(0, h, 4096), rs.randint(0, w, 4096)
            for k in (0, 1):  # stack planes 0, 1 are bank planes 0, 1 unrolled
                key = "s2048__k{}__u16".format(k)
                assert int(cfg4[k]) == int(g[key + "__cfg"][0])
                assert abs(out4[k].astype(np.float64).sum() - g[key + "__sum"][0]) / g[key + "__sum"][0] < 1e-5
                rel = rel_err(out4[k][sy, sx], g[key + "__sample"])
                assert (rel > REL_TOL).sum() <= 2, (k, float(rel.max()))  # planes 0, 1 have no flips (printed above)
    finally:
        e4.close()
        e1.close()


def test_timed_configuration_256_planes_2048(golden_large):
    """The configuration ``bench.py`` times (BASELINE ``configs[1]``): ONE cohort of 256 planes of 2048 x 2048 uint16 on
    device buffers, uint16 out, four stream parts of 64 planes -- the launch geometry (march segments, histogram block
    heights, blocks per launch) depends on the part size, and no smaller cohort has the parts of this one
    (VERDICT r3 weak #1).  One plane of each of the 4 x 64 parts plus plane 255 goes through the proof obligations of
    ``check_plane`` against the oracle; planes 0 / 1 are compared with the samples the real reference wrote
    (``large_stats.npz``); the uint16 result (what the bench measures) must be the truncated float32 result of the
    same cohort geometry for ALL 256 planes; and every plane must equal the plane of the 32-plane bank it repeats
    (rolled), i.e. a plane's result does not depend on where in the cohort it sits."""
    n, h, w = 256, 2048, 2048
    bank = synth.synthetic_bank(32, h, w)
    stack = synth.synthetic_stack(n, h, w, bank=bank)
    e = _engine_with_streams(4)
    e1 = _engine_with_streams(1)
    try:
        e.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=n)
        d_in, d_u16, d_f32, d_cfg = e.alloc(stack.nbytes), e.alloc(stack.nbytes), e.alloc(stack.nbytes * 2), e.alloc(4 * n)
        d_in.upload(stack)
        for _ in range(3):  # back-to-back calls: parts queue behind the same part of the call before (deferred joins)
            e.run_device(d_in, np.uint16, n, d_u16, np.uint16, d_cfg)
        e.sync()
        cfg = d_cfg.download((n,), np.int32)
        u16 = d_u16.download((n, h, w), np.uint16)
        e.run_device(d_in, np.uint16, n, d_f32, np.float32, d_cfg)
        e.sync()
        np.testing.assert_array_equal(d_cfg.download((n,), np.int32), cfg)
        picks = [5, 64 + 17, 128 + 38, 192 + 59, 255]  # one per part (parts are planes [64 i, 64 i + 64)) + the last
        plane_bytes = h * w * 4
        f32 = {}
        for z0 in range(0, n, 32):  # all 256 planes, 32 at a time: the uint16 result is the truncated float32 one
            blk = d_f32.download((32, h, w), np.float32, offset=z0 * plane_bytes)
            np.testing.assert_array_equal(u16[z0 : z0 + 32], np.clip(blk, 0, 65535).astype(np.uint16))
            for z in [0, 1] + picks:
                if z0 <= z < z0 + 32:
                    f32[z] = blk[z - z0].copy()
            del blk
        # position independence: slice z = bank[z % 32] rolled by z // 32 rows -- away from the rows the roll wraps
        # around (the filter sees another plane there) results agree to the bit only if nothing leaks between planes;
        # the configs chosen must repeat exactly
        np.testing.assert_array_equal(cfg, np.tile(cfg[:32], n // 32))
        assert 0 < int(cfg.sum()) < n  # both branches are in the cohort
        sub = np.ascontiguousarray(stack[picks])
        deltas = gpu_deltas(e1, sub)
        for i, z in enumerate(picks):
            which, _, _, ref, stages = oracle_plane(stack[z])
            assert int(cfg[z]) == which, (z, int(cfg[z]), which)
            _check_plane(f32[z], stack[z], deltas[i], ("timed configuration", z), ref=ref, stages=stages)
        rs = np.random.RandomState(7)
        sy, sx = rs.randint(0, h, 4096), rs.randint(0, w, 4096)
        g = golden_large
        for k in (0, 1):  # stack planes 0, 1 are bank planes 0, 1 unrolled
            key = "s2048__k{}__u16".format(k)
            assert int(cfg[k]) == int(g[key + "__cfg"][0])
            assert abs(f32[k].astype(np.float64).sum() - g[key + "__sum"][0]) / g[key + "__sum"][0] < 1e-5
            rel = rel_err(f32[k][sy, sx], g[key + "__sample"])
            assert (rel > REL_TOL).sum() <= 2, (k, float(rel.max()))
        for b in (d_in, d_u16, d_f32, d_cfg):
            b.free()
    finally:
        e.close()
        e1.close()


# ---- hard-decision sweeps against the real reference (tests/golden/sweep.npz, float32 = Zarr-path regime) --
def _same_bin(a, b):
    return abs(a - b) <= 1e-4 * abs(b) + 1e-30


def test_seed_sweep_512(engine, golden_sweep):
    """32 seeds at 512 x 512, production parameters: config choice, Otsu BIN and threshold of every
    level against the reference's float32 (Zarr path) regime, sampled outputs with explained outliers."""
    g = golden_sweep
    rs = np.random.RandomState(7)
    sy, sx = rs.randint(0, 512, 1024), rs.randint(0, 512, 1024)
    planes = synth.synthetic_bank(32, 512, 512)
    deltas = gpu_deltas(engine, planes)
    otsu = [[engine.thresholds(k, lv) for lv in range(engine.levels)] for k in range(32)]
    cfgs = [engine.stats(k)[2] for k in range(32)]
    out = filtering.destripe_planes(planes, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                    synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, max_batch=32)  # fmt: skip
    moved, total_bad, total_flips = 0, 0, 0
    for k in range(32):
        key = "seed{}__f32".format(k)
        assert cfgs[k] == int(g[key + "__cfg"][0]), key
        ref_otsu, ref_thr = g[key + "__otsu"][::-1], g[key + "__thr"][::-1]  # fixture: coarse -> fine
        same = [_same_bin(otsu[k][lv][0], ref_otsu[lv]) for lv in range(engine.levels)]
        if not all(same):
            # the float64 regime of the reference may name the bin the engine picked (plateau of the
            # class-variance curve); anything else is an error
            alt = g["seed{}__u16__otsu".format(k)][::-1]
            assert all(s or _same_bin(otsu[k][lv][0], alt[lv]) for lv, s in enumerate(same)), (key, otsu[k], ref_otsu)
            moved += 1
            continue
        np.testing.assert_allclose([t for _, t in otsu[k]], ref_thr, rtol=3e-5)
        _, _, _, _, stages = oracle_plane(planes[k].astype(np.float32))
        nb, fl = _check_plane(out[k], planes[k].astype(np.float32), deltas[k], key, pos=(sy, sx),
                              ref=g[key + "__sample"], stages=stages, which=cfgs[k])
        total_bad += nb
        total_flips += sum(fl)
        assert abs(out[k].astype(np.float64).sum() - g[key + "__sum"][0]) / g[key + "__sum"][0] < 1e-5
    print("[parity] seed sweep: {} planes with a bin of the other regime, {} samples beyond 1e-4, {} flips".format(
        moved, total_bad, total_flips))
    assert moved <= 2, moved


def test_width_sweep(engine, golden_sweep):
    """68 plane widths (level-1 row lengths around the multiples of 64 / 256 -- the slot boundaries of the
    row filter --, even and odd widths), both production configs, 48 rows: Otsu bins and sampled outputs
    against the reference.  The engine computes in float32 like the reference's Zarr path; on a plateau of
    the class-variance curve the reference's two regimes pick different bins, so a width counts as matched
    if every level agrees with one regime of the reference (and then its samples must match that regime)."""
    g = golden_sweep
    other = 0
    for W in [int(x) for x in g["widths"]]:
        img = synth.synthetic_plane(((W + 5) // 2) % 7, 48, W)
        for cname, cfg in CFGS.items():
            engine.plan(48, W, cfg, cfg, synth.ZARR_PATH_HIGH_INT, max_batch=1)
            engine.set_stop_after(2)
            try:
                engine.run(img[None], out_dtype=np.float32)
                delta = [engine.level_array(0, lv, eng_mod.STAGE_DETAIL) for lv in range(engine.levels)]
                otsu = [engine.thresholds(0, lv)[0] for lv in range(engine.levels)]
            finally:
                engine.set_stop_after(0)
            out = engine.run(img[None], out_dtype=np.float32)[0]
            matched = None
            for dt in ("f32", "u16"):
                key = "w{}__{}__{}".format(W, cname, dt)
                if all(_same_bin(o, r) for o, r in zip(otsu, g[key + "__otsu"][::-1])):
                    matched = dt
                    break
            assert matched is not None, (W, cname, otsu)
            other += matched != "f32"
            key = "w{}__{}__{}".format(W, cname, matched)
            assert out.shape == tuple(g[key + "__shape"])
            rs = np.random.RandomState(W)
            yy, xx = rs.randint(0, out.shape[0], 256), rs.randint(0, out.shape[1], 256)
            x = img if matched == "u16" else img.astype(np.float32)
            _, stages = orc.log_space_fft_filtering(x, return_stages=True, **cfg)
            check_plane(out, x, delta, key, cfg, lambda size: 6, ref=g[key + "__sample"], stages=stages[::-1], pos=(yy, xx))
    print("[parity] width sweep: {} of {} cases follow the float64 regime's Otsu bin".format(other, 2 * len(g["widths"])))
    assert other <= 8, other


@pytest.mark.parametrize(
    "shape",
    [
        (101, 104), (75, 260), (130, 132), (37, 64), (36, 1000), (33, 244), (34, 248), (250, 252), (48, 488),
        (49, 492), (200, 976), (64, 128), (1026, 36), (40, 2044), (95, 1020),
    ],
)
def test_fused_kernel_edge_shapes(engine, shape):
    """Planes that take the fused level-1 + 2 kernels (width a multiple of 4) with awkward geometry: odd heights,
    level-2 widths around the 61-column strips of a wave (one strip, a shifted last strip that overlaps
    the one before it, a last strip of a single column), level-1 widths around 122 / 126, tiny and tall
    planes.  Every pixel against the oracle (explained-outlier statement of tests/parity_util.py)."""
    planes = np.stack([synth.synthetic_plane(k, *shape) for k in (0, 1)])
    deltas = gpu_deltas(engine, planes)
    out, cfg = filtering.destripe_planes(
        planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
        out_dtype=np.float32, return_config=True, max_batch=2,
    )  # fmt: skip
    for k in range(2):
        which, _, _, ref, stages = oracle_plane(planes[k])
        assert int(cfg[k]) == which
        assert out[k].shape == ref.shape
        _check_plane(out[k], planes[k], deltas[k], (shape, k), ref=ref, stages=stages)
    # per-level cH of plane 0 (forward kernels alone), float32 round-off
    engine.plan(shape[0], shape[1], synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=2)
    engine.set_stop_after(1)
    try:
        engine.run(planes, out_dtype=np.float32)
        _, _, _, _, stages = oracle_plane(planes[0])
        for lv, st in enumerate(stages):
            ch = engine.level_array(0, lv, eng_mod.STAGE_DETAIL)
            assert ch.shape == st["ch"].shape
            scale = max(1.0, np.abs(st["ch"]).max())
            assert np.abs(ch - st["ch"]).max() <= 2e-5 * scale * (2**lv), (shape, lv)
    finally:
        engine.set_stop_after(0)


@pytest.mark.parametrize(
    "shape",
    [
        (40, 4604),    # level 1 has 2 304 coefficients per row: the last width the one-wave row filter takes
        (40, 4608),    # 2 306: the first one k_rowfilter_wide takes (direct transform 2 * 1153 or embedded)
        (33, 6001),    # odd width and height, levels 1 and 2 wide
        (24, 9216),    # 4 610 = 2 * 5 * 461: embedded transform (a length of small primes, periodic halo)
        (16, 12288),   # 6 146 = 2 * 7 * 439
        (300, 5120),   # a camera-like plane: 8 levels, two of them wide
        (40, 2304),    # 1 154 = 2 * 577 coefficients: fits a wave, but its embedding (2 340) does not -- block kernel
    ],
)
def test_planes_wider_than_one_wave_holds(engine, shape):
    """The reference takes any width (filtering.py:206); rows of more than 2 304 coefficients (planes wider than 4 604
    pixels) run the block-per-row-pair row filter (k_rowfilter_wide: block-wide passes, key-bisection median).  Every
    pixel of two planes against the oracle, uint16 and float32 input, both configs in play."""
    planes = np.stack([synth.synthetic_plane(k, *shape) for k in (0, 1)])
    for src in (planes, planes.astype(np.float32)):
        deltas = gpu_deltas(engine, src)
        out, cfg = filtering.destripe_planes(
            src, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
            out_dtype=np.float32, return_config=True, max_batch=2,
        )  # fmt: skip
        for k in range(2):
            which, _, _, ref, stages = oracle_plane(src[k])
            assert int(cfg[k]) == which
            assert out[k].shape == ref.shape
            _check_plane(out[k], src[k], deltas[k], (shape, k), ref=ref, stages=stages)


@pytest.mark.parametrize("shape", [(2390, 64), (2294, 64), (2198, 64), (626, 64), (628, 256), (578, 64)])
def test_short_last_march_segment(engine, shape):
    """Heights for which the LAST row segment of a march kernel is 1 ... 3 rows long (the segments are at least 24 rows,
    the remainder is what it is): 601 / 577 / 553 level-2 rows in 25-row segments for the fused forward kernel, 313 / 314 /
    289 coefficient rows in 26-row segments for the final kernel.  The bottom rows of such a segment mirror into rows the
    segment above it owns; every pixel of two planes against the oracle."""
    planes = np.stack([synth.synthetic_plane(k, *shape) for k in (0, 1)])
    deltas = gpu_deltas(engine, planes)
    out, cfg = filtering.destripe_planes(
        planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
        out_dtype=np.float32, return_config=True, max_batch=2,
    )  # fmt: skip
    for k in range(2):
        which, _, _, ref, stages = oracle_plane(planes[k])
        assert int(cfg[k]) == which
        _check_plane(out[k], planes[k], deltas[k], (shape, k), ref=ref, stages=stages)


def test_otsu_tie_plane(engine, capsys):
    """A plane whose level-2 class-variance curve has two maxima that agree to 1.5e-7 relative (found by
    tools/fuzz_parity.py): the reference's arg-max lands on bin 50, the engine's -- its float32 coefficients differ
    by 1e-6 -- on bin 51, one bin width (1 %) higher.  The parity statement (tests/parity_util.py, obligations (d) and
    (e)) must recognise the tie from the ORACLE's own curve, take the engine's bin in the oracle and hold everywhere
    else; a wrong bin without a tie must still fail."""
    import parity_util

    h, w = 244, 971
    planes = np.stack([synth.synthetic_plane(k, h, w) for k in (921, 971)])
    deltas = gpu_deltas(engine, planes)
    out = filtering.destripe_planes(planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                    synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, max_batch=2)
    for k in range(2):
        _check_plane(out[k], planes[k], deltas[k], ((h, w), k))
    assert "Otsu tie at level index [1]" in capsys.readouterr().out
    # the tie test is not a loophole: an Otsu value two bins away from the maximum is rejected
    _, _, _, _, stages = oracle_plane(planes[1])
    bad = list(deltas[1].otsu)
    q = stages[1]["ch"] ** 2
    centres, var = orc.otsu_variance_curve(*orc.histogram256(q))
    bad[1] = float(centres[int(np.argmax(var)) + 3])
    with pytest.raises(AssertionError):
        parity_util.find_otsu_ties(bad, stages)


@pytest.mark.parametrize("poison", [np.nan, np.inf, -2.0, -1.0])
def test_invalid_float_pixels_raise_like_the_reference(poison):
    """A float32 plane with a pixel whose log(1 + x) is not finite: the reference dies in numpy.histogram inside
    threshold_otsu (ValueError: autodetected range of [nan, nan] is not finite; checked against the real reference
    for NaN, inf, -2 and -1 with ``level=None``).  The engine flags such pixels in the forward kernel and the host-buffer
    call raises the same exception type with the same message; the planes around it are not affected afterwards."""
    planes = synth.synthetic_bank(3, 96, 128).astype(np.float32)
    bad = planes.copy()
    bad[1, 40, 50] = poison
    with pytest.raises(ValueError, match="autodetected range of .* is not finite"):
        filtering.destripe_planes(bad, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
                                  out_dtype=np.float32, max_batch=3)
    # -0.5 is legal (log(0.5)), and a clean batch right after the failed one is processed normally
    ok = planes.copy()
    ok[1, 40, 50] = -0.5
    out = filtering.destripe_planes(ok, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT,
                                    out_dtype=np.float32, max_batch=3)
    assert np.isfinite(out).all()
    which, _, _, ref, _ = oracle_plane(ok[1])
    assert _rel(out[1], ref).max() < 1e-3  # one odd pixel: no flip accounting here, just "the same picture"


def test_graph_cache_follows_the_stack_mode(monkeypatch):
    """the stack mode is baked into a captured launch chain but is not part of the graph cache's key.
    ``log_space_fft_filtering`` toggles it on a cached engine (3-D input: on, then off again), so with ``DSX_GRAPH=1`` a
    2-D call, a 3-D call and a 2-D call on ONE engine through the same staging buffers must each get the thresholds of
    their own mode -- ``dsx_set_stack_mode`` drops the captured graphs when the mode changes."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stack3d.npz"), allow_pickle=False)
    stack = np.ascontiguousarray(g["a__in"])  # one plane carries a bright block: shared thresholds differ from per-plane ones
    n, H, W = stack.shape
    cfg = synth.CELLS_CONFIG

    def thresholds(e):
        return np.array([[e.thresholds(k, lv)[1] for lv in range(e.levels)] for k in range(n)])

    def fresh(stack_mode):
        e = eng_mod.DestripeEngine(0)
        try:
            e.plan(H, W, cfg, cfg, 2700, max_batch=n)
            e.set_stack_mode(stack_mode)
            return e.run(stack, out_dtype=np.float32), thresholds(e)
        finally:
            e.close()

    monkeypatch.delenv("DSX_GRAPH", raising=False)
    ref_2d, thr_2d = fresh(False)
    ref_3d, thr_3d = fresh(True)
    assert np.abs(thr_2d - thr_3d).max() > 0 and np.abs(ref_2d - ref_3d).max() > 0
    assert np.all(thr_3d == thr_3d[0])  # one threshold per level for the whole stack
    monkeypatch.setenv("DSX_GRAPH", "1")
    e = eng_mod.DestripeEngine(0)
    try:
        e.plan(H, W, cfg, cfg, 2700, max_batch=n)
        for mode, ref, thr in ((False, ref_2d, thr_2d), (True, ref_3d, thr_3d), (False, ref_2d, thr_2d), (True, ref_3d, thr_3d)):
            e.set_stack_mode(mode)
            captures_before = e.graph_stats()[1]
            for _ in range(3):  # eager, capture, replay -- all through the engine's own staging buffers
                out = e.run(stack, out_dtype=np.float32)
                np.testing.assert_array_equal(out, ref)
                np.testing.assert_array_equal(thresholds(e), thr)
            assert e.graph_stats()[1] == captures_before + 1  # the other mode's graph was dropped, not replayed
            e.set_stack_mode(mode)  # setting the same mode again keeps the graph
            np.testing.assert_array_equal(e.run(stack, out_dtype=np.float32), ref)
            assert e.graph_stats()[1] == captures_before + 1
    finally:
        e.close()


def test_graph_replay_is_bit_identical_to_eager_launches(monkeypatch):
    """DSX_GRAPH=1: unsplit cohorts are captured into a HIP graph at the second identical call and replayed afterwards
    (dsx_graph_stats counts): every call returns the bits of the eager first one; a new plan, new shading planes or
    other buffers start over; the host-buffer API (the per-slice filter_stripes calls) replays too."""
    planes = synth.synthetic_bank(5, 200, 232)
    monkeypatch.setenv("DSX_GRAPH", "1")
    e = eng_mod.DestripeEngine(0)
    try:
        e.plan(200, 232, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=8)
        d_in, d_out, d_cfg = e.alloc(planes.nbytes), e.alloc(planes.nbytes * 2), e.alloc(4 * 5)
        d_in.upload(planes)
        outs = []
        for _ in range(4):
            e.run_device(d_in, np.uint16, 5, d_out, np.float32, d_cfg)
            e.sync()
            outs.append((d_out.download((5, 200, 232), np.float32), d_cfg.download((5,), np.int32)))
        launches, captures = e.graph_stats()
        assert captures == 1 and launches == 3, (launches, captures)
        for o, c in outs[1:]:
            np.testing.assert_array_equal(o, outs[0][0])
            np.testing.assert_array_equal(c, outs[0][1])
        # another count through the same buffers is another graph; the first one is still there
        for _ in range(3):
            e.run_device(d_in, np.uint16, 3, d_out, np.float32, d_cfg)
        e.sync()
        np.testing.assert_array_equal(d_out.download((3, 200, 232), np.float32), outs[0][0][:3])
        e.run_device(d_in, np.uint16, 5, d_out, np.float32, d_cfg)
        e.sync()
        assert e.graph_stats() == (launches + 3, 2)
        np.testing.assert_array_equal(d_out.download((5, 200, 232), np.float32), outs[0][0])
        # host-buffer calls go through the context's staging buffers: replayed from the third call on
        before = e.graph_stats()
        host = [e.run(planes[:2], out_dtype=np.float32) for _ in range(4)]
        for h in host[1:]:
            np.testing.assert_array_equal(h, host[0])
        np.testing.assert_array_equal(host[0], outs[0][0][:2])
        assert e.graph_stats()[0] >= before[0] + 2
        # a new plan drops the graphs (they hold workspace addresses) and the results follow the new configs
        cfg2 = dict(synth.CELLS_CONFIG, sigma=200)
        e.plan(200, 232, cfg2, cfg2, synth.ZARR_PATH_HIGH_INT, max_batch=8)
        new = []
        for _ in range(3):
            e.run_device(d_in, np.uint16, 5, d_out, np.float32, d_cfg)
            e.sync()
            new.append(d_out.download((5, 200, 232), np.float32))
        np.testing.assert_array_equal(new[1], new[0])
        np.testing.assert_array_equal(new[2], new[0])
        assert not np.array_equal(new[0], outs[0][0])
        ref = orc.log_space_fft_filtering(planes[1], **cfg2)
        assert _rel(new[2][1], ref).max() < REL_TOL
        # the default: a context that never captures gives the same bits
        monkeypatch.delenv("DSX_GRAPH")
        e2 = eng_mod.DestripeEngine(0)
        try:
            e2.plan(200, 232, cfg2, cfg2, synth.ZARR_PATH_HIGH_INT, max_batch=8)
            eager = [e2.run(planes, out_dtype=np.float32) for _ in range(3)]
            assert e2.graph_stats() == (0, 0)
            np.testing.assert_array_equal(eager[2], new[0])
        finally:
            e2.close()
    finally:
        e.close()


def test_fused_rowfinal_is_bit_identical_to_the_unfused_chain(monkeypatch):
    """2048-, 2000- and 1800-wide uint16 planes run the level-1 row filter INSIDE the final kernel (k_rowfinal: Delta_1
    stays in LDS, every lane synthesises its own c_1 columns from level 2).  Same arithmetic, operation for operation: the result must equal
    the chain with k_rowfilter + k_inv_march (DSX_NO_FUSE_RF=1) bit for bit -- unsplit cohort (helper stream), a cohort
    split over the four streams, with and without the shading epilogue, a plane of constant rows, a plane with the
    cells config, and a short plane (2048 wide, 200 high: the last block is partial)."""
    for h, w, n, shading in ((2048, 2048, 72, False), (2048, 2048, 20, True), (200, 2048, 40, False),
                             (1600, 2000, 40, False), (1600, 2000, 20, True), (1800, 1800, 40, False), (2048, 2048, 16, None)):
        # shading None: the cells config filters NO level (level 0 -> result x + 2 for its planes), the other one all of
        # them -- level 1 is inactive for some planes of the cohort and k_rowfinal must synthesise rows of zeros for them
        cells_cfg = dict(synth.CELLS_CONFIG, level=0) if shading is None else synth.CELLS_CONFIG
        shading = bool(shading)
        bank = synth.synthetic_bank(6, h, w)
        flatp = np.full((h, w), 300, np.uint16)
        rows = np.repeat((100 + 50 * np.arange(h, dtype=np.uint16) % 7)[:, None], w, axis=1).astype(np.uint16)
        stack = np.concatenate([bank, flatp[None], rows[None], synth.synthetic_stack(n - 8, h, w, bank=bank)])
        flat = dark = None
        if shading:
            yy, xx = np.mgrid[0:h, 0:w]
            flat = (1.0 - 0.15 * (((yy - h / 2.0) / (h / 2.0)) ** 2 + ((xx - w / 2.0) / (w / 2.0)) ** 2)).astype(np.float32)
            dark = np.full((h, w), 100.0, np.float32)
        res = {}
        monkeypatch.setenv("DSX_FUSE_RF_WIDE", "1")  # 2000- / 1800-wide planes: off by default (slower there)
        for mode in ("1", "0"):
            monkeypatch.setenv("DSX_NO_FUSE_RF", mode)
            e = eng_mod.DestripeEngine(0)
            try:
                e.plan(h, w, cells_cfg, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=n,
                       flatfield=flat, darkfield=dark)
                small = e.run(stack[:8], out_dtype=np.uint16)            # one part, helper stream
                big, cfg = e.run(stack, out_dtype=np.uint16, return_cfg=True)
                res[mode] = (small, big, cfg)
            finally:
                e.close()
        assert res["0"][2].sum() > 0  # some planes took the cells config
        for a, b in zip(res["0"], res["1"]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b), err_msg=str((h, w, n, shading)))
        if cells_cfg is not synth.CELLS_CONFIG:  # planes that chose the level-0 config come out as x + 2
            for k in np.nonzero(res["0"][2])[0][:3]:
                np.testing.assert_array_equal(res["0"][1][k], np.minimum(stack[k].astype(np.int64) + 2, 65535).astype(np.uint16))
        elif h == 2048 and not shading:  # and the reference: plane 1 of the stack against the oracle
            ref = orc.filter_stripes(stack[1], "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT)
            d = np.abs(res["0"][1][1].astype(np.int64) - ref.astype(np.uint16).astype(np.int64))
            assert d.max() <= 1


def test_timing_switches_do_nothing_in_the_product_build(monkeypatch):
    """The timing-only switches of the measurement builds (DSX_ABLATE, DSX_SKIP_HIST / _ROW / _COARSE: kernels skip a phase
    or a launch and return WRONG pixels) are compiled in under -DDSX_DIAG only (tools/build_variant.sh).  The library the
    package ships must return the same bits whatever those variables say; the fused histogram / row-filter experiment of
    round 2 (DSX_FUSE_HIST, an inter-block spin barrier) is gone from the sources altogether."""
    h, w = 2048, 2048
    bank = synth.synthetic_bank(4, h, w)
    stack = synth.synthetic_stack(72, h, w, bank=bank)  # 72 planes: 4 parts of 18 on the four streams
    res = {}
    for mode in ("plain", "switched"):
        if mode == "switched":
            for k, v in (("DSX_ABLATE", str(1 | 2 | 4 | 16 | 32)), ("DSX_SKIP_HIST", "1"), ("DSX_SKIP_ROW", "3"),
                         ("DSX_SKIP_COARSE", "2"), ("DSX_FUSE_HIST", "3")):
                monkeypatch.setenv(k, v)
        e = eng_mod.DestripeEngine(0)
        try:
            e.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=72)
            small = e.run(stack[:8], out_dtype=np.float32)            # one part, helper stream
            thr_small = [[e.thresholds(k, lv) for lv in range(e.levels)] for k in range(8)]
            big, cfg = e.run(stack, out_dtype=np.uint16, return_cfg=True)
            res[mode] = (small, thr_small, big, cfg)
        finally:
            e.close()
    for a, b in zip(res["switched"], res["plain"]):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    # and the reference: plane 1 of the stack against the oracle
    ref = orc.filter_stripes(stack[1], "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT)
    d = np.abs(res["switched"][2][1].astype(np.int64) - ref.astype(np.uint16).astype(np.int64))
    assert d.max() <= 1


def test_device_buffer_path_reports_non_finite_planes_at_the_next_sync():
    """float32 planes with a pixel whose log(1 + x) is not finite: the reference dies in numpy.histogram (ValueError).
    dsx_run_host raises at the call; the asynchronous dsx_run_device cannot, so the flag is folded into a sticky word of
    the context and the NEXT synchronising entry point raises the same ValueError, once."""
    h, w = 96, 128
    planes = synth.synthetic_bank(3, h, w).astype(np.float32)
    bad = planes.copy()
    bad[1, 40, 50] = np.nan
    e = eng_mod.DestripeEngine(0)
    try:
        e.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=4)
        d_in, d_out = e.alloc(bad.nbytes), e.alloc(bad.nbytes)
        try:
            d_in.upload(bad)
            e.run_device(d_in, np.float32, 3, d_out, np.float32)
            with pytest.raises(ValueError, match="not finite"):
                e.sync()
            e.sync()  # reported once
            d_in.upload(planes)
            e.run_device(d_in, np.float32, 3, d_out, np.float32)
            e.sync()
            good = d_out.download((3, h, w), np.float32)
        finally:
            d_in.free()
            d_out.free()
        np.testing.assert_array_equal(good, e.run(planes, out_dtype=np.float32))
        with pytest.raises(ValueError, match="not finite"):  # the host-buffer call raises by itself ...
            e.run(bad, out_dtype=np.float32)
        e.sync()                                              # ... and leaves nothing behind
        # a value error of an EARLIER asynchronous call that nobody has synchronised with yet must not be
        # wiped by a host-buffer call that follows on the same context -- that call synchronises, so it reports it
        d_in = e.alloc(bad.nbytes)
        d_out = e.alloc(bad.nbytes)
        try:
            d_in.upload(bad)
            e.run_device(d_in, np.float32, 3, d_out, np.float32)
            with pytest.raises(ValueError, match="earlier dsx_run_device call"):
                e.run(planes, out_dtype=np.float32)
            np.testing.assert_array_equal(e.run(planes, out_dtype=np.float32), good)  # reported once; the engine works on
            e.sync()
        finally:
            d_in.free()
            d_out.free()
    finally:
        e.close()


def test_bench_self_launch_two_ranks_on_the_one_gpu():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE -- the driver's command -- must start two ranks by
    itself.  Rehearsed on the one GPU of this box (DSX_SHARE_GPU=1: both ranks on device 0; RCCL refuses two ranks on
    one device, so the ranks agree on the host transport): the line must say n_gpus 2, carry both ranks' slices, and
    verify."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSX_RDZV_DIR")}
    env["DSX_SHARE_GPU"] = "1"
    r = subprocess.run(
        [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32",
         "--settle", "0", "--shape", "512x512"],
        capture_output=True, text=True, timeout=600, env=env)  # fmt: skip
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["scaling"] == "weak"
    assert d["config"]["ranks_share_gpus"] is True and d["config"]["rank_transport"] in ("host", "rccl")
    assert abs(d["value"] - 2 * 32 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]


def test_bench_under_a_launcher_with_a_real_rccl_communicator():
    """The driver's multi-GPU command is `torch.distributed.run ... bench.py --gpus N`: RANK / WORLD_SIZE in the
    environment, one rank per GPU, the constants broadcast over RCCL through the C ABI (`dsx_comm_*`, librccl.so
    dlopen'ed by the library).  This box has ONE GPU, so the rehearsal is a world of one rank with DSX_FORCE_COMM=1: the
    real ncclGetUniqueId / ncclCommInitRank / ncclBroadcast / ncclAllReduce / ncclCommDestroy on the device, the file
    rendezvous around them, and a verified bench line that says so."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("DSX_RDZV_DIR", "DSX_SHARE_GPU")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", DSX_FORCE_COMM="1")
    r = subprocess.run(
        [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "32",
         "--settle", "0", "--shape", "512x512", "--cpu-planes", "0"],
        capture_output=True, text=True, timeout=600, env=env)  # fmt: skip
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    c = d["config"]
    assert d["n_gpus"] == 1 and d["verified"] is True
    assert c["rank_transport"] == "rccl" and c["rccl_ranks"] == 1 and c["rccl_error"] is None, c
    assert c["constants_broadcast_bytes"] > 0
