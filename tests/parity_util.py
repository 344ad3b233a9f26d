"""Shared helpers of the GPU parity tests: the tolerance statement and its proof obligations.

Tolerance (BASELINE.json north star): every pixel within 1e-4 relative of the reference -- EXCEPT
pixels that lie under a *hard decision that fell the other way*.  The path thresholds every cH
coefficient (``mask = |cH| > thr``, filtering.py:195); a coefficient whose magnitude is within
float32 round-off of the threshold can land on the other side than in the reference (whose own
float32 and float64 regimes differ from each other in the same way).  Such a flip at level ``l``,
coefficient row ``i`` changes Delta_l along that whole row (the row median and the low-pass are
per-row operators) and reaches, through ``l`` db3 synthesis steps, the result rows

    [2^l i - 4 (2^l - 1),  2^l i + 2^l - 1]      (all columns).

So the tests (a) obtain the engine's masks (``dsx_set_stop_after(2)``: Delta == 0 marks a masked
coefficient), (b) compare them with the oracle's, (c) require EVERY pixel beyond 1e-4 to lie in the
row band of a flipped coefficient, (d) cap what a flip may do (``OUTLIER_CAP``), and (e) bound and
print the number of flips per level.  A localized kernel bug (an edge strip, a tail slot) is not
under a flipped coefficient and fails (c).
"""

import numpy as np

from aind_smartspim_destripe_amd import engine as eng_mod
from aind_smartspim_destripe_amd import synth
from oracle import destripe_oracle as orc

REL_TOL = 1e-4      # north star tolerance
OUTLIER_CAP = 5e-2  # no pixel, flipped footprint or not, may be further off than this


def rel_err(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - b) / np.abs(b)


def oracle_plane(img, high_int=synth.ZARR_PATH_HIGH_INT, cells=None, nocells=None):
    """(config index, fore mean, back mean, output, stages fine -> coarse) of the CPU oracle."""
    cells = cells or synth.CELLS_CONFIG
    nocells = nocells or synth.NO_CELLS_CONFIG
    which, fore, back = orc.select_config(img, nocells, cells, high_int)
    cfg = cells if which else nocells
    out, stages = orc.log_space_fft_filtering(img, return_stages=True, **cfg)
    return which, fore, back, out, stages[::-1]


def gpu_deltas(engine, planes, high_int=synth.ZARR_PATH_HIGH_INT, max_batch=None):
    """Delta_l of every plane and level from an engine run stopped after the row filter."""
    n, h, w = planes.shape
    engine.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, high_int, max_batch=max_batch or n)
    engine.set_stop_after(2)
    try:
        engine.run(planes, out_dtype=np.float32)
        return [[engine.level_array(k, lv, eng_mod.STAGE_DETAIL) for lv in range(engine.levels)] for k in range(n)]
    finally:
        engine.set_stop_after(0)


def flip_rows(deltas, stages, out_h):
    """Result rows reachable from coefficients whose mask bit differs between engine and oracle.

    ``deltas``: Delta per level of one plane (fine -> coarse); ``stages``: oracle stages (fine -> coarse).
    Returns (bool[out_h] rows that may legitimately exceed REL_TOL, flips per level).
    """
    rows = np.zeros(out_h, dtype=bool)
    counts = []
    for lv, (delta, st) in enumerate(zip(deltas, stages)):
        mask_ref = np.abs(st["ch"]) > st["threshold"]
        mask_gpu = delta == 0.0  # an unmasked coefficient has Delta == 0 only by coincidence
        flipped = np.nonzero((mask_gpu != mask_ref).any(axis=1))[0]
        counts.append(int((mask_gpu != mask_ref).sum()))
        s = 1 << (lv + 1)
        for i in flipped:
            lo, hi = s * int(i) - 4 * (s - 1), s * int(i) + s - 1
            rows[max(lo, 0) : min(hi, out_h - 1) + 1] = True
    return rows, counts


def assert_close_explained(out, ref, rows_ok, what, pos=None, max_outliers=None):
    """Every pixel within REL_TOL, except under flipped coefficients (``rows_ok``), and none beyond the cap.

    ``pos = (sy, sx)``: ``out`` / ``ref`` are samples at those positions of the plane.
    Returns the number of pixels beyond REL_TOL (all of them explained).
    """
    rel = rel_err(out, ref)
    bad = rel > REL_TOL
    n_bad = int(bad.sum())
    assert float(rel.max()) < OUTLIER_CAP, (what, "outlier beyond the cap", float(rel.max()))
    if n_bad:
        bad_rows = np.nonzero(bad)[0] if pos is None else np.asarray(pos[0])[bad]
        unexplained = ~rows_ok[bad_rows]
        assert not unexplained.any(), (
            what, "pixels beyond 1e-4 outside every flipped footprint", int(unexplained.sum()),
            np.unique(bad_rows[unexplained])[:10].tolist(), float(rel.max()))
    if max_outliers is not None:
        assert n_bad <= max_outliers, (what, n_bad, max_outliers)
    assert float(np.median(rel)) < 1e-5, (what, float(np.median(rel)))
    return n_bad
