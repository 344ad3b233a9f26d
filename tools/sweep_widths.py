#!/usr/bin/env python3
"""Parity sweep over plane widths whose level-1 row length sits around the slot boundaries of the row filter
(multiples of 64 and 256): engine vs the NumPy oracle, both production configs.  Diagnosis / pre-release check."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_smartspim_destripe_amd import filtering, synth  # noqa: E402
from oracle import destripe_oracle as orc  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 48
worst = 0.0
for w1 in [63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319, 320, 383, 384, 385, 511, 512, 513, 575, 767,
           768, 769, 1001, 1023, 1024, 1025, 1026, 1027, 1151, 1152, 1153, 1500]:
    for W in (2 * w1 - 4, 2 * w1 - 5):
        img = synth.synthetic_plane(w1 % 7, H, W)
        for name, cfg in (("cells", synth.CELLS_CONFIG), ("nocells", synth.NO_CELLS_CONFIG)):
            out = filtering.log_space_fft_filtering(img, **cfg)
            # The engine computes in float32, like the reference's Zarr path (float32 planes,
            # zarr_destriper.py:1049); its TIFF path (uint16 planes, all float64) can pick another Otsu bin
            # on a plateau of the class-variance curve (empty bins) -- the two regimes of the reference
            # differ from each other there, so the sweep accepts agreement with either.
            rel, bad = None, None
            for regime in (np.float32, np.uint16):
                ref = orc.log_space_fft_filtering(img.astype(regime), **cfg)
                assert out.shape == ref.shape, (W, out.shape, ref.shape)
                r = np.abs(out - ref) / np.abs(ref)
                b = int((r > 1e-4).sum())
                if bad is None or b < bad:
                    rel, bad = r, b
            worst = max(worst, float(np.median(rel)))
            status = "ok" if bad <= max(300, int(1e-3 * rel.size)) and np.median(rel) < 1e-5 else "FAIL"  # a threshold flip costs one footprint (level-1 flips spread along the row)
            if status != "ok" or name == "cells" and W % 2 == 0:
                print("w1=%4d W=%4d %-7s max %.2e median %.2e bad %d %s" % (w1, W, name, rel.max(), np.median(rel), bad, status))
            assert status == "ok", (w1, W, name, bad)
print("sweep ok; worst median rel", worst)
