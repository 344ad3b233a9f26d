#!/usr/bin/env python3
"""Randomised parity run: engine (through the C ABI) against the CPU oracle on random plane shapes, input dtypes, batch
sizes and output dtypes, with the statement of tests/parity_util.py (every pixel within 1e-4 once the counted
near-threshold mask decisions are forced into the oracle).  usage: python tools/fuzz_parity.py [cases] [seed] [wavelets | geometry]
(third argument "wavelets": a random PyWavelets wavelet per case -- 2 ... 102 taps -- through the tap-count-generic level
kernels, on small planes down to a few pixels, the oracle running on the same filter bank;
"geometry": every case ALSO draws the launch-geometry switches of the library -- DSX_SEG_MIN_ROWS, DSX_MARCH_WAVES,
DSX_FWD_WPB / DSX_INV_WPB, DSX_ROW_WPB, DSX_HIST_ROWS, DSX_STREAMS, DSX_HELPER, DSX_NO_QUANT, DSX_NO_ROW_MULTI,
DSX_ROW_MULTI_ALONE, DSX_NO_PAIR, DSX_NO_FUSE, DSX_NO_FUSE_INV, DSX_NO_FUSE_RF, DSX_FUSE_RF_WIDE, DSX_NO_PIPELINE, DSX_GRAPH,
DSX_PRIO: each moves
segment boundaries, block heights or the launch chain, none may move a result -- on a fresh context per case (dsx_init
reads them per context), with big multi-stream batches three times as often)
Shapes favour the awkward ones: widths around multiples of 4 / 8 / 122 / 244 / 256 (strip and lane-pair boundaries of
the march kernels), odd heights, planes of one strip and of many."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from aind_smartspim_destripe_amd import engine as eng_mod, filtering, synth, wavelets
from oracle import destripe_oracle as orc
from parity_util import check_plane, gpu_deltas, oracle_plane



GEOMETRY = {  # switch -> values a case draws from (None = unset: the library's default)
    "DSX_SEG_MIN_ROWS": [None, 2, 3, 4, 5, 8, 24],
    "DSX_MARCH_WAVES": [None, 64, 256, 1024, 4096, 16384],
    "DSX_FWD_WPB": [None, 4, 8],
    "DSX_INV_WPB": [None, 4, 8],
    "DSX_ROW_WPB": [None, 4, 5, 6, 7, 8],
    "DSX_HIST_ROWS": [None, 1, 32, 128, 256],
    "DSX_STREAMS": [None, 1, 2, 3, 4],
    "DSX_HELPER": [None, 0, 1],
    "DSX_NO_QUANT": [None, 1],
    "DSX_NO_ROW_MULTI": [None, 1],
    "DSX_ROW_MULTI_ALONE": [None, 0, 1000],
    "DSX_NO_PAIR": [None, None, 1],
    "DSX_NO_FUSE": [None, None, None, 1],
    "DSX_NO_FUSE_INV": [None, None, None, 1],
    "DSX_NO_FUSE_RF": [None, None, 1],
    "DSX_NO_PIPELINE": [None, None, 1],
    "DSX_GRAPH": [None, None, None, 1],
    "DSX_FUSE_RF_WIDE": [None, None, 1],           # k_rowfinal for the embedded plans of 2000- / 1800-wide planes (off by default)
    "DSX_PRIO": [None, None, None, "0,-1,0,-1", "-1,0,1,0"],  # HIP stream priorities of the sub-cohort streams
}


def draw_geometry(rng, apply=True):
    """Sets / unsets the switches of GEOMETRY in os.environ; returns the drawn non-default ones.  ``apply=False`` ("geometry-dry":
    the same random stream, the library's defaults) tells a switch's doing from the plane's when a case fails."""
    drawn = {}
    for name, values in GEOMETRY.items():
        v = values[int(rng.integers(0, len(values)))]
        if not apply:
            os.environ.pop(name, None)
            continue
        if v is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(v)
            drawn[name] = v
    return drawn


def run(cases=40, seed=2026, eng=None, any_wavelet=False, geometry=False):
    """``cases`` random cases; raises AssertionError on the first plane that violates the parity statement.
    Returns (planes checked, planes that needed the float32 regime of the reference)."""
    rng = np.random.default_rng(seed)
    own = eng is None
    max_flips = lambda size: max(3, int(2e-5 * size))
    eng = eng_mod.DestripeEngine(0) if own else eng
    t0 = time.time(); worst = 0.0; n_regime_retries = 0; n_planes = 0; n_batched = 0
    saved_env = {k: os.environ.get(k) for k in GEOMETRY}
    for c in range(cases):
        drawn = None
        if geometry:  # a fresh context per case: dsx_init reads the switches
            filtering.release_engines()
            if own and eng is not None:
                eng.close()
            drawn = draw_geometry(rng, apply=geometry != "dry")
            eng = eng_mod.DestripeEngine(0)
            own = True
        anchor = int(rng.choice([64, 122, 128, 244, 256, 488, 512, 732, 976, 1220, 1708, 2048]))
        w = max(40, anchor + int(rng.integers(-9, 10)))
        if rng.random() < 0.6: w = (w + 3) & ~3          # fused kernels need a multiple of 4
        if rng.random() < 0.3: w = (w + 7) & ~7          # lane-pair I/O of the final kernel: multiple of 8
        h = int(rng.integers(40, 700))
        if geometry and rng.random() < 0.22:             # the hot widths: k_rowfinal and the compile-time FFT plans
            w, h = int(rng.choice([2048, 2048, 2000, 1800])), int(rng.integers(64, 420)) & ~1
        if not any_wavelet and rng.random() < 0.03:      # rows longer than one wave holds (k_rowfilter_wide), few rows
            w = int(rng.choice([2304, 4606, 4608, 5120, 6001, 7000, 9216])) + int(rng.integers(0, 3))
            h = int(rng.integers(12, 90))
        if not any_wavelet and rng.random() < 0.03:      # tall and narrow: many row segments, a last one of any length
            h = int(rng.integers(2100, 4300))
            w = (int(rng.integers(40, 76)) + 3) & ~3
        name = "db3"
        if any_wavelet:
            # (dmey is refused; rbio3.1 amplifies float32 round-off beyond 1e-4 at depth -- the reference's own float32
            # and float64 regimes differ by 1e-3 there -- and has its own test with its own bound, tests/test_wavelets.py)
            name = str(rng.choice([x for x in wavelets.wavelist() if x not in ("dmey", "rbio3.1")]))
            w, h = int(rng.integers(6, 420)), int(rng.integers(6, 300))
        n = int(rng.integers(1, 5))
        as_f32 = rng.random() < 0.3
        # random filter parameters: decomposition depth, low-pass width, threshold cap, fg/bg decision level
        cells = {"wavelet": name, "level": [None, None, 1, 2, 3, 5][int(rng.integers(0, 6))],
                 "sigma": float(rng.choice([16, 64, 100, 250])), "max_threshold": float(rng.choice([0.5, 3, 12]))}
        nocells = {"wavelet": name, "level": [None, None, 1, 2, 4][int(rng.integers(0, 5))],
                   "sigma": float(rng.choice([32, 128, 512])), "max_threshold": float(rng.choice([1, 12, 100]))}
        high_int = int(rng.choice([100, 160, 2500]))
        # the oracle takes the filter bank itself (it knows db3 by name only)
        ocells, onocells = (dict(c, wavelet=wavelets.filter_bank(name)) if any_wavelet else c for c in (cells, nocells))
        planes = np.stack([synth.synthetic_plane(int(rng.integers(0, 1000)), h, w) for _ in range(n)])
        src = planes.astype(np.float32) if as_f32 else planes
        try:
            deltas = gpu_deltas(eng, src, high_int=high_int, cells=cells, nocells=nocells)
        except ValueError as e:
            if "odd plane" in str(e):  # documented limit: one config with levels, one without, on an odd plane
                continue
            raise
        out, cfg = filtering.destripe_planes(src, "X_0_Y_0", nocells, cells, None,
                                             high_int, out_dtype=np.float32, return_config=True, max_batch=n)
        for k in range(n):
            # The engine computes in float32 like the reference's Zarr path (float32 planes, zarr_destriper.py:1049); for
            # uint16 input the reference's TIFF path runs in float64 and can pick the neighbouring Otsu bin on a plateau
            # of the class-variance curve -- the two regimes of the reference differ from each other there, so a plane must
            # satisfy the statement against ONE of them (the float64 regime is tried first).
            err = None
            for regime in ((np.float32,) if as_f32 else (np.uint16, np.float32)):
                img = src[k].astype(regime)
                which, _, _, ref, stages = oracle_plane(img, high_int=high_int, cells=ocells, nocells=onocells)
                # (no decomposition level at all -- long filters on small planes: the result is image + 2 whatever the
                # config, the statistic is not computed and cfg_used reads 0)
                assert int(cfg[k]) == which or eng.levels == 0, (h, w, k)
                cfgd = ocells if which else onocells
                try:
                    check_plane(out[k], img, deltas[k], ((h, w), np.dtype(regime).name, k, name, drawn), cfgd, max_flips, ref=ref, stages=stages)
                    err = None
                    break
                except AssertionError as e:
                    err = e
                    n_regime_retries += 1
            if err is not None:
                # keep the failing plane: the CPU side (oracle, histograms) can then be studied without a GPU
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d.npz" % (seed, c)), plane=src[k], out=out[k],
                                        cfg=int(cfg[k]), cells=repr(cells), nocells=repr(nocells), high_int=high_int,
                                        otsu=np.array(deltas[k].otsu, dtype=np.float64), drawn=repr(drawn),
                                        **{"ch%d" % lv: np.asarray(a) for lv, a in enumerate(deltas[k].ch)})
                except Exception as save_err:  # noqa: BLE001
                    print("could not save the failing case:", save_err)
                raise err
            n_planes += 1
        # now and then: the same planes many times over, split into cohorts and sub-cohort streams -- every copy of a
        # plane must come out bit-identical to the small run above, wherever it sits in the batch
        if rng.random() < (0.4 if geometry else 0.12) and h * w <= (420 * 2048 if geometry else 300 * 600):
            reps = int(rng.integers(33, 80))
            idx = rng.integers(0, n, size=reps)
            big = src[idx]
            mb = int(rng.choice([16, 32, 48, reps]))
            out_big = filtering.destripe_planes(big, "X_0_Y_0", nocells, cells, None, high_int, out_dtype=np.float32,
                                                max_batch=mb)
            assert np.array_equal(out_big, out[idx]), ("batch position dependence", (h, w), reps, mb, drawn)
            n_batched += 1
        # uint16 result path (truncation): within one count of the float result
        out16 = filtering.destripe_planes(src, "X_0_Y_0", nocells, cells, None, high_int, out_dtype=np.uint16, max_batch=n)
        d = np.abs(out16.astype(np.float64) - np.clip(np.floor(out.astype(np.float64)), 0, 65535))
        assert d.max() <= 1.0, ((h, w), float(d.max()), drawn)
        worst = max(worst, float(d.max()))
    if geometry:
        filtering.release_engines()
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print("fuzz: %d cases passed in %.0f s (uint16 vs floor(float32) result: max difference %.0f count; %d planes matched the float32 regime of the reference only; %d big-batch bit-identity checks)" % (cases, time.time() - t0, worst, n_regime_retries, n_batched))
    if own:
        eng.close()
    return n_planes, n_regime_retries


if __name__ == "__main__":
    import warnings
    warnings.simplefilter("ignore")  # "level too high" for long filters on small planes
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 2026,
        any_wavelet=len(sys.argv) > 3 and sys.argv[3] == "wavelets",
        geometry=("dry" if sys.argv[3] == "geometry-dry" else True) if len(sys.argv) > 3 and sys.argv[3].startswith("geometry") else False)
