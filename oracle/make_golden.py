"""Generate tests/golden/*.npz by running the REAL reference in this container.

Run (only here; /root/reference does not exist on the GPU box and is never copied)::

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/code \
        /opt/conda/bin/python3.9 -W ignore oracle/make_golden.py

The reference module ``aind_smartspim_destripe.filtering`` is imported from /root/reference/code
and called as is.  To record per-level internals (Otsu input/result, row medians) without touching
the reference, ``filters.threshold_otsu`` and ``np.median`` are wrapped for the duration of a call.
Only inputs/outputs (data) are written; interpreter + library versions go into every file.
"""

import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(REPO, "tests", "golden")

from aind_smartspim_destripe import filtering as ref  # noqa: E402  (the reference itself)

_spec = importlib.util.spec_from_file_location(
    "dsx_synth", os.path.join(REPO, "aind_smartspim_destripe_amd", "synth.py")
)
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


def versions():
    import pywt
    import scipy
    import skimage

    return json.dumps(
        {
            "python": sys.version.split()[0],
            "numpy": np.__version__,
            "scipy": scipy.__version__,
            "pywt": pywt.__version__,
            "skimage": skimage.__version__,
            "reference": "AllenNeuralDynamics/aind-smartspim-destripe @ 2025-05-23 (/root/reference)",
        }
    )


class Recorder:
    """Wraps skimage Otsu and np.median while the reference runs, to expose per-level internals."""

    def __init__(self):
        self.otsu_in = []
        self.otsu_out = []
        self.medians = []

    def __enter__(self):
        self._otsu = ref.filters.threshold_otsu
        self._median = np.median

        def otsu(image, *a, **k):
            r = self._otsu(image, *a, **k)
            self.otsu_in.append(np.array(image))
            self.otsu_out.append(float(r))
            return r

        def median(arr, *a, **k):
            r = self._median(arr, *a, **k)
            self.medians.append(np.array(r))
            return r

        ref.filters.threshold_otsu = otsu
        np.median = median
        return self

    def __exit__(self, *exc):
        ref.filters.threshold_otsu = self._otsu
        np.median = self._median


def run_lsf(image, cfg):
    """log_space_fft_filtering + recorded internals (levels ordered coarse -> fine)."""
    with Recorder() as rec:
        out = ref.log_space_fft_filtering(input_image=image, **cfg)
    thr = [min(cfg["max_threshold"], float(np.sqrt(o))) for o in rec.otsu_out]
    mask_counts = [int((np.sqrt(q) > t).sum()) for q, t in zip(rec.otsu_in, thr)]
    return out, rec.otsu_out, thr, mask_counts, rec.medians


def make_small():
    """Full float64 outputs on small planes, both dtype regimes."""
    d = {"versions": versions()}
    cases = []
    cfgs = {"cells": dict(synth.CELLS_CONFIG), "nocells": dict(synth.NO_CELLS_CONFIG)}

    def add(name, img, cfg_name, level, dtypes=("u16", "f32")):
        cfg = dict(cfgs[cfg_name])
        cfg["level"] = level
        for dt in dtypes:
            x = img if dt == "u16" else img.astype(np.float32)
            out, otsu, thr, mc, med = run_lsf(x, cfg)
            key = "{}__{}__L{}__{}".format(name, cfg_name, "max" if level is None else level, dt)
            d[key + "__out"] = out
            d[key + "__otsu"] = np.array(otsu, dtype=np.float64)
            d[key + "__thr"] = np.array(thr, dtype=np.float64)
            d[key + "__maskcount"] = np.array(mc, dtype=np.int64)
            cases.append(key)

    for name, (h, w, k) in {
        "p64": (64, 64, 0),
        "p128x96": (128, 96, 1),
        "p101x103": (101, 103, 4),
    }.items():
        img = synth.synthetic_plane(k, h, w)
        d[name + "__in"] = img
        for cfg_name in ("cells", "nocells"):
            for level in (1, None):
                add(name, img, cfg_name, level)
    img = synth.synthetic_plane(0, 64, 64)
    add("p64", img, "cells", 0)
    add("p64", img, "cells", 2, dtypes=("u16",))

    img = synth.synthetic_plane(8, 256, 256)
    d["p256__in"] = img
    add("p256", img, "cells", None, dtypes=("u16",))
    add("p256", img, "nocells", None, dtypes=("u16",))

    # the reference's own test input (code/tests/test_filtering.py:156): 100x100 float32 ramp, level=1
    ramp = np.tile(np.linspace(1, 100, 100), (100, 1)).astype(np.float32)
    out = ref.log_space_fft_filtering(ramp, "db3", 1, 64, 4)
    d["ramp100__L1__out"] = out
    out = ref.log_space_fft_filtering(ramp, "db3", None, 64, 4)
    d["ramp100__Lmax__out"] = out
    # tiny plane with an over-deep level (code/tests/test_filtering.py:171-180 shape)
    tiny = (np.random.RandomState(5).rand(4, 4) * 500).astype(np.float32)
    d["tiny4__in"] = tiny
    d["tiny4__L1__out"] = ref.log_space_fft_filtering(tiny, wavelet="db3", level=1, sigma=64, max_threshold=4)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLDEN, "small_full.npz"), **d)
    print("small_full:", len(cases), "cases")


def make_large():
    """Per-level internals + sampled outputs of filter_stripes at the BASELINE shapes."""
    d = {"versions": versions()}
    shapes = {"s512": (512, 512), "s1800": (1800, 1800), "s1600x2000": (1600, 2000), "s2048": (2048, 2048)}
    cases = []
    for name, (h, w) in shapes.items():
        for k in (0, 1):
            img = synth.synthetic_plane(k, h, w)
            d["{}__k{}__insum".format(name, k)] = np.array([img.astype(np.uint64).sum()], dtype=np.uint64)
            rs = np.random.RandomState(7)
            sy = rs.randint(0, h, 4096)
            sx = rs.randint(0, w, 4096)
            for dt in ("u16", "f32"):
                x = img if dt == "u16" else img.astype(np.float32)
                fore, back, _ = ref.get_foreground_background_mean(x)
                with Recorder() as rec:
                    out = ref.filter_stripes(
                        image=x,
                        input_tile_path="X_0_Y_0",
                        no_cells_config=dict(synth.NO_CELLS_CONFIG),
                        cells_config=dict(synth.CELLS_CONFIG),
                        shadow_correction=None,
                        microscope_high_int=synth.ZARR_PATH_HIGH_INT,
                    )
                use_cells = bool(fore > back and fore > synth.ZARR_PATH_HIGH_INT)
                cfg = synth.CELLS_CONFIG if use_cells else synth.NO_CELLS_CONFIG
                thr = [min(cfg["max_threshold"], float(np.sqrt(o))) for o in rec.otsu_out]
                mc = [int((np.sqrt(q) > t).sum()) for q, t in zip(rec.otsu_in, thr)]
                key = "{}__k{}__{}".format(name, k, dt)
                d[key + "__cfg"] = np.array([1 if use_cells else 0])
                d[key + "__means"] = np.array([float(fore), float(back)])
                d[key + "__otsu"] = np.array(rec.otsu_out)
                d[key + "__thr"] = np.array(thr)
                d[key + "__maskcount"] = np.array(mc, dtype=np.int64)
                d[key + "__medians"] = np.concatenate([m.ravel() for m in rec.medians])
                d[key + "__sample"] = out[sy, sx]
                d[key + "__sum"] = np.array([out.sum()])
                d[key + "__shape"] = np.array(out.shape)
                cases.append(key)
                print(key, "cfg", int(use_cells), "levels", len(thr))
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLDEN, "large_stats.npz"), **d)


def make_misc():
    """fg/bg statistic, notch/gaussian known answers, flatfield known answers, shaded filter_stripes."""
    d = {"versions": versions()}
    # (1) fg/bg statistic on every uint16 value, both dtypes
    allv = np.arange(65536, dtype=np.uint16)
    for dt, x in (("u16", allv), ("f32", allv.astype(np.float32))):
        fore, back, mask = ref.get_foreground_background_mean(x)
        d["fgbg_all__{}__means".format(dt)] = np.array([float(fore), float(back)])
        d["fgbg_all__{}__mask".format(dt)] = np.packbits(mask.astype(np.uint8))
    # (2) the float16 mask decision for EVERY float16 bit pattern (finite, non-NaN ones are used)
    h = np.arange(65536, dtype=np.uint16).view(np.float16)
    with np.errstate(all="ignore"):
        f = ref.foreground_fraction(h, 400, 20)
        m = f > 0.3
    d["f16_mask_table"] = np.packbits(m.astype(np.uint8))
    # (3) fractional float32 pixels around the cut-off
    fr = np.linspace(380.0, 388.0, 1025).astype(np.float32)
    fore, back, mask = ref.get_foreground_background_mean(fr)
    d["fgbg_frac__in"] = fr
    d["fgbg_frac__mask"] = mask.astype(np.uint8)
    d["fgbg_frac__means"] = np.array([float(fore), float(back)])
    # (4) notch / gaussian_filter
    d["notch_5_1"] = ref.notch(5, 1.0)
    d["notch_1026_32"] = ref.notch(1026, 32.0625)
    d["gauss_3x5_1"] = ref.gaussian_filter((3, 5), 1.0)
    # (5) flatfield known answer of the reference test + a float case
    d["flat_kat"] = ref.flatfield_correction(
        np.array([[[10, 20], [30, 40]]]), np.array([[[2, 2], [2, 2]]]), np.array([[[1, 1], [1, 1]]])
    )
    rs = np.random.RandomState(11)
    img = rs.rand(48, 40) * 70000.0
    flat = 0.5 + rs.rand(48, 40)
    dark = rs.rand(60, 50) * 300.0
    d["flat_f__img"], d["flat_f__flat"], d["flat_f__dark"] = img, flat, dark
    d["flat_f__out"] = ref.flatfield_correction(img, flat, dark)
    # (6) filter_stripes with retrospective shading on a 128x96 plane (uint16 result)
    for k in (0, 1):
        img = synth.synthetic_plane(k, 128, 96)
        yy, xx = np.mgrid[0:128, 0:96]
        flat = (1.0 - 0.3 * (((yy - 64) / 64.0) ** 2 + ((xx - 48) / 48.0) ** 2) / 2.0).astype(np.float32)
        dark = np.full((140, 110), 100.0, dtype=np.float32)
        sc = {"retrospective": True, "flatfield": flat, "darkfield": dark, "tile_config": {}}
        out = ref.filter_stripes(
            image=img,
            input_tile_path="X_0_Y_0",
            no_cells_config=dict(synth.NO_CELLS_CONFIG),
            cells_config=dict(synth.CELLS_CONFIG),
            shadow_correction=sc,
            microscope_high_int=synth.ZARR_PATH_HIGH_INT,
        )
        d["shade__k{}__in".format(k)] = img
        d["shade__k{}__out".format(k)] = out
        if k == 0:
            d["shade__flat"], d["shade__dark"] = flat, dark
    np.savez_compressed(os.path.join(GOLDEN, "misc.npz"), **d)
    print("misc done")


SWEEP_W1 = [63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319, 320, 383, 384, 385, 511, 512, 513, 575, 767,
            768, 769, 1001, 1023, 1024, 1025, 1026, 1027, 1151, 1152, 1153, 1500]  # fmt: skip
SWEEP_H = 48
SWEEP_SEEDS = 32


def make_sweep():
    """Hard-decision sweeps (Otsu returns a bin centre, the mask is a hard threshold), both dtype regimes:

    (1) 32 seeds at 512 x 512 through filter_stripes (production parameters): per-level Otsu value,
        threshold, mask count, 1024 sampled output pixels, plane sum, chosen config;
    (2) 68 plane widths (level-1 row lengths around the multiples of 64 / 256, even and odd widths) at
        48 rows through log_space_fft_filtering with both production configs: per-level Otsu value and
        256 sampled output pixels.
    """
    d = {"versions": versions()}
    rs = np.random.RandomState(7)
    sy, sx = rs.randint(0, 512, 1024), rs.randint(0, 512, 1024)
    for k in range(SWEEP_SEEDS):
        img = synth.synthetic_plane(k, 512, 512)
        for dt in ("u16", "f32"):
            x = img if dt == "u16" else img.astype(np.float32)
            fore, back, _ = ref.get_foreground_background_mean(x)
            with Recorder() as rec:
                out = ref.filter_stripes(
                    image=x,
                    input_tile_path="X_0_Y_0",
                    no_cells_config=dict(synth.NO_CELLS_CONFIG),
                    cells_config=dict(synth.CELLS_CONFIG),
                    shadow_correction=None,
                    microscope_high_int=synth.ZARR_PATH_HIGH_INT,
                )
            use_cells = bool(fore > back and fore > synth.ZARR_PATH_HIGH_INT)
            cfg = synth.CELLS_CONFIG if use_cells else synth.NO_CELLS_CONFIG
            thr = [min(cfg["max_threshold"], float(np.sqrt(o))) for o in rec.otsu_out]
            key = "seed{}__{}".format(k, dt)
            d[key + "__cfg"] = np.array([1 if use_cells else 0])
            d[key + "__otsu"] = np.array(rec.otsu_out)  # coarse -> fine
            d[key + "__thr"] = np.array(thr)
            d[key + "__maskcount"] = np.array([int((np.sqrt(q) > t).sum()) for q, t in zip(rec.otsu_in, thr)])
            d[key + "__sample"] = out[sy, sx]
            d[key + "__sum"] = np.array([out.sum()])
    print("sweep: seeds done")
    cfgs = {"cells": dict(synth.CELLS_CONFIG), "nocells": dict(synth.NO_CELLS_CONFIG)}
    for w1 in SWEEP_W1:
        for W in (2 * w1 - 4, 2 * w1 - 5):
            img = synth.synthetic_plane(w1 % 7, SWEEP_H, W)
            for cname, cfg in cfgs.items():
                for dt in ("u16", "f32"):
                    x = img if dt == "u16" else img.astype(np.float32)
                    out, otsu, thr, mc, _ = run_lsf(x, cfg)
                    rs = np.random.RandomState(W)
                    yy, xx = rs.randint(0, out.shape[0], 256), rs.randint(0, out.shape[1], 256)
                    key = "w{}__{}__{}".format(W, cname, dt)
                    d[key + "__otsu"] = np.array(otsu)
                    d[key + "__sample"] = out[yy, xx]
                    d[key + "__shape"] = np.array(out.shape)
    d["widths"] = np.array(sorted({2 * w1 - 4 for w1 in SWEEP_W1} | {2 * w1 - 5 for w1 in SWEEP_W1}))
    np.savez_compressed(os.path.join(GOLDEN, "sweep.npz"), **d)
    print("sweep done")


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    only = sys.argv[1:]
    for name, fn in (("small", make_small), ("misc", make_misc), ("large", make_large), ("sweep", make_sweep)):
        if not only or name in only:
            fn()
