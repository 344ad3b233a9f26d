#!/usr/bin/env python3
"""Degenerate planes through the engine and the oracle under the parity statement of tests/parity_util.py: all zero,
saturated, two flat halves, one hot pixel, constant rows / columns, checkerboard, a ramp, pure noise at 0..3 counts,
a bright block on zero background.  (Constant coefficient levels take Otsu's early-out, zero thresholds mask nothing
or everything, medians sit inside the zero spike ...)  usage: python tools/fuzz_patterns.py"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from aind_smartspim_destripe_amd import engine as eng_mod, filtering, synth
from parity_util import check_plane, gpu_deltas, oracle_plane


def patterns(h, w, rng):
    yy, xx = np.mgrid[0:h, 0:w]
    p = {}
    p["zeros"] = np.zeros((h, w), np.uint16)
    p["saturated"] = np.full((h, w), 65535, np.uint16)
    p["halves"] = np.where(xx < w // 2, 100, 3000).astype(np.uint16)
    p["halves_rows"] = np.where(yy < h // 2, 100, 3000).astype(np.uint16)
    hot = np.full((h, w), 120, np.uint16); hot[h // 3, w // 5] = 65535
    p["hot_pixel"] = hot
    p["const_rows"] = (100 + 37 * (yy % 11)).astype(np.uint16)
    p["const_cols"] = (100 + 37 * (xx % 13)).astype(np.uint16)
    p["checker"] = (200 + 150 * ((yy + xx) & 1)).astype(np.uint16)
    p["ramp"] = ((yy * 65535) // max(h - 1, 1)).astype(np.uint16)
    p["low_noise"] = rng.integers(0, 4, (h, w)).astype(np.uint16)
    blk = np.zeros((h, w), np.uint16); blk[h // 4 : h // 2, w // 4 : w // 2] = 5000
    p["block"] = blk
    stripes = (300 + 40 * np.sin(yy / 3.0) + rng.poisson(20, (h, w))).astype(np.uint16)
    p["stripes"] = stripes
    return p


def run(shapes=((96, 128), (203, 260), (260, 1028)), seed=3):
    rng = np.random.default_rng(seed)
    eng = eng_mod.DestripeEngine(0)
    max_flips = lambda size: max(3, int(2e-5 * size))
    n = 0
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for h, w in shapes:
            pats = patterns(h, w, rng)
            names = list(pats)
            planes = np.stack([pats[k] for k in names])
            deltas = gpu_deltas(eng, planes)
            out, cfg = filtering.destripe_planes(planes, "X_0_Y_0", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None,
                                                 synth.ZARR_PATH_HIGH_INT, out_dtype=np.float32, return_config=True,
                                                 max_batch=len(names))
            for k, name in enumerate(names):
                err = None
                for regime in (np.uint16, np.float32):
                    img = planes[k].astype(regime)
                    which, _, _, ref, stages = oracle_plane(img)
                    if not np.all(np.isfinite(ref)):
                        # the reference itself produces non-finite values here: the engine must do so in the same pixels
                        assert np.array_equal(np.isfinite(out[k]), np.isfinite(ref)), (name, (h, w), "finite pattern")
                        print("[pattern] %s %s: reference is non-finite in %d pixels, engine in the same" % (name, (h, w), int((~np.isfinite(ref)).sum())))
                        err = None
                        break
                    assert int(cfg[k]) == which, (name, (h, w))
                    cfgd = synth.CELLS_CONFIG if which else synth.NO_CELLS_CONFIG
                    try:
                        check_plane(out[k], img, deltas[k], (name, (h, w), np.dtype(regime).name), cfgd, max_flips, ref=ref, stages=stages)
                        err = None
                        break
                    except AssertionError as e:
                        err = e
                if err is not None:
                    raise err
                n += 1
    eng.close()
    print("patterns: %d planes passed" % n)
    return n


if __name__ == "__main__":
    run()
