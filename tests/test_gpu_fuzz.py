"""A short randomised parity run inside the GPU suite: 30 random cases of tools/fuzz_parity.py (plane shapes around
the strip / lane-pair boundaries of the march kernels, uint16 / float32 input, 1-4 planes, random filter parameters
per config) under the full statement of tests/parity_util.py.  The seed is fixed: the test is deterministic; longer
runs with other seeds are a tool (profiles/r2_fuzz_parity.log: 2 492 planes)."""

import os
import sys
import warnings

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_fuzz_parity_30_cases():
    import fuzz_parity

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "level too high" for tiny planes, as pywt warns in the reference
        planes, _ = fuzz_parity.run(cases=30, seed=99)
    assert planes >= 30


def test_degenerate_planes():
    """All-zero, saturated, two-valued, one hot pixel, constant rows / columns, checkerboard, ramp, 0..3-count noise,
    a block on zero background (tools/fuzz_patterns.py) at three shapes: constant coefficient levels (Otsu's early-out,
    thresholds of 0, levels that are pure round-off noise) against the oracle under the same statement."""
    import fuzz_patterns

    assert fuzz_patterns.run() == 36
