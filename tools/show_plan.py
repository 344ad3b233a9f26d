#!/usr/bin/env python3
"""Print the row-filter plan (FFT length and halo per level) of the library in use (DSX_LIB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_smartspim_destripe_amd import engine, synth  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W = int(sys.argv[2]) if len(sys.argv) > 2 else H
e = engine.DestripeEngine(0)
i = e.plan(H, W, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, 2500, max_batch=4)
print(os.environ.get("DSX_LIB", "default"), [(i.level_w[k], i.fft_len[k], i.fft_halo[k]) for k in range(i.levels)])
