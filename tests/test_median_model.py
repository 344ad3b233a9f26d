"""The exact row median of the row-filter kernels (value-domain bracketing, csrc/dsx_kernels.h) as a NumPy model
(tools/median_model.py): the round-3 loop -- Illinois false position, lo / hi adjacency only tested on the fallback path,
a finished row stepping on while its partner runs -- must return np.median's value bit for bit on adversarial rows, and
must keep the bracket invariant C(lo) <= k < C(hi) through extra steps.  CPU only; the kernel itself is held to the
oracle's medians by the GPU parity tests (stage row filter, golden vectors, fuzz)."""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from median_model import kernel_median, kernel_median3  # noqa: E402


def _rows(seed):
    rs = np.random.RandomState(seed)
    for n in (12, 20, 36, 68, 132, 260, 515, 1026, 1027, 1002, 902):
        for kind in range(12):
            thr = np.float32(10 ** rs.uniform(-3, 1))
            if kind == 0: x = rs.randn(n) * thr / 3
            elif kind == 1: x = np.full(n, rs.randn() * thr / 3)                      # one value
            elif kind == 2: x = rs.choice(rs.randn(3) * thr / 4, n)                   # three values
            elif kind == 3: x = rs.randn(n) * thr / 3; x[rs.rand(n) < rs.uniform(0, 0.9)] = 0   # masked entries
            elif kind == 4: x = np.abs(rs.randn(n)) * thr / 3
            elif kind == 5: x = -np.abs(rs.randn(n)) * thr / 3
            elif kind == 6: x = rs.randn(n) * thr * 1e-6
            elif kind == 7: x = np.round(rs.randn(n) * 3) * thr / 10                  # heavy ties
            elif kind == 8: x = rs.standard_cauchy(n) * thr / 50                      # heavy tails (clipped at the threshold)
            elif kind == 9: x = rs.randn(n) * thr / 3 + thr / 2
            elif kind == 10: x = np.full(n, thr)                                      # everything at the upper end
            else: x = np.where(rs.rand(n) < 0.5, thr, -thr)                           # two spikes at the ends
            yield np.clip(x, -thr, thr).astype(np.float32), thr


@pytest.mark.parametrize("extra", [0, 5, 40])
def test_round3_median_loop_is_exact(extra):
    counts = []
    for x, thr in _rows(11 + extra):
        m, n = kernel_median3(x, thr, extra)
        assert m == np.float32(np.median(x)), (x.size, float(thr), extra)
        counts.append(n)
    assert max(counts) <= 2 + 256


def test_round2_median_loop_is_exact_and_not_cheaper():
    c2, c3 = [], []
    for x, thr in _rows(5):
        m2, n2 = kernel_median(x, thr)
        m3, n3 = kernel_median3(x, thr)
        assert m2 == m3 == np.float32(np.median(x))
        c2.append(n2)
        c3.append(n3)
    assert np.mean(c3) <= np.mean(c2) * 1.05
